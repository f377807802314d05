"""GPU parity of the Wan2.1 T2V rollout path against the CPU oracle (oracle/wan_ref.py).  Through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.bfloat16().float()


@pytest.fixture(scope="module")
def wn():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import wan
    return wan


def _setup(wn, cfg_o, seed=55):
    from oracle import wan_ref as R
    sd = {k: _bf(v) for k, v in R.make_synthetic_state_dict(cfg_o, seed).items()}
    cfg = wn.WanConfig(num_layers=cfg_o.num_layers, num_attention_heads=cfg_o.num_attention_heads, ffn_dim=cfg_o.ffn_dim,
                       text_dim=cfg_o.text_dim)
    return sd, cfg


@pytest.mark.parametrize("B,T,h,w,Nt,n_cfg", [(2, 3, 8, 12, 9, 1), (1, 2, 6, 10, 64, 2), (2, 1, 16, 16, 17, 2)])
def test_wan_forward_matches_oracle(wn, B, T, h, w, Nt, n_cfg):
    """Self-attention with 3-D RoPE and across-head RMSNorm, cross-attention on the cached text K / V^T (S_q != S_kv, ragged text
    length), modulation tables, un-patchify; n_cfg == 2 = one forward over [negative, positive]."""
    from oracle import wan_ref as R
    cfg_o = R.tiny_config()
    sd, cfg = _setup(wn, cfg_o)
    eng = wn.WanEngine(cfg)
    eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    eng.ready()
    g = torch.Generator().manual_seed(T * h + w + Nt)
    x = torch.randn(B, 16, T, h, w, generator=g).half()
    pe = _bf(torch.randn(B, Nt, cfg_o.text_dim, generator=g))
    ne = _bf(torch.randn(B, Nt, cfg_o.text_dim, generator=g))
    t = torch.tensor([874.0])
    plan = eng.plan(B, n_cfg, T, h, w, Nt, 1)
    if n_cfg == 2:
        got = plan.transformer_forward(x.cuda(), t, ne.cuda(), pe.cuda()).float().cpu()
        ref = torch.cat([R.wan_forward(sd, cfg_o, x.float(), t.expand(B), ne), R.wan_forward(sd, cfg_o, x.float(), t.expand(B), pe)])
    else:
        got = plan.transformer_forward(x.cuda(), t, pe.cuda()).float().cpu()
        ref = R.wan_forward(sd, cfg_o, x.float(), t.expand(B), pe)
    assert got.shape == ref.shape
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel
    eng.close()


def test_wan_forward_with_48_latent_channels_matches_oracle(wn):
    """Wan2.2-TI2V-5B's latent geometry (48 channels: patch-embedding K = 192, proj_out N = 192) in text-to-video use, where the adapter's
    per-token timesteps (`expand_timesteps`, reference models/wan/wan2_t2v.py:502-504) are all equal to t: the engine's scalar-timestep
    forward (the plugin's choice, flow_factory_plugin.py) vs the oracle."""
    import dataclasses
    from oracle import wan_ref as R
    cfg_o = dataclasses.replace(R.tiny_config(), in_channels=48, out_channels=48)
    sd = {k: _bf(v) for k, v in R.make_synthetic_state_dict(cfg_o, 56).items()}
    cfg = wn.WanConfig(in_channels=48, out_channels=48, num_layers=cfg_o.num_layers, num_attention_heads=cfg_o.num_attention_heads,
                       ffn_dim=cfg_o.ffn_dim, text_dim=cfg_o.text_dim)
    eng = wn.WanEngine(cfg)
    eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    eng.ready()
    B, T, h, w, Nt = 2, 3, 8, 10, 11
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 48, T, h, w, generator=g).bfloat16()
    pe = _bf(torch.randn(B, Nt, cfg_o.text_dim, generator=g))
    ne = _bf(torch.randn(B, Nt, cfg_o.text_dim, generator=g))
    t = torch.tensor([611.0])
    got = eng.plan(B, 2, T, h, w, Nt, 1).transformer_forward(x.cuda(), t, ne.cuda(), pe.cuda()).float().cpu()
    ref = torch.cat([R.wan_forward(sd, cfg_o, x.float(), t.expand(B), ne), R.wan_forward(sd, cfg_o, x.float(), t.expand(B), pe)])
    assert got.shape == ref.shape == (2 * B, 48, T, h, w)
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel
    eng.close()


@pytest.mark.parametrize("B,S,H", [(2, 100, 2), (1, 777, 12), (2, 130, 40)])
def test_norm_rope_measures_the_largest_stored_row_norm_per_batch_and_head(wn, B, S, H):
    """`mi355_op_norm_rope_full` with `max2`: the atomic maximum, per (batch, head), of the squared norm of every row AS STORED (bf16) -- the
    number the self-attention's data-dependent score bound is built from (a value too SMALL would let exp2 overflow in the static kernel).
    Checked against torch on the operator's own output; one row per (b, h) is made an outlier."""
    import ctypes as C
    from mi355_flow import _lib
    from mi355_flow.engine import _ptr, _stream
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + S + H)
    D = H * 128
    M, S_pad = B * S, (S + 63) // 64 * 64
    src = torch.randn(M, D, device="cuda", generator=g)
    for b in range(B):
        for h in range(H):
            src[b * S + (7 * h + 3 * b) % S, h * 128:(h + 1) * 128] *= 3.0 + h % 5
    src = src.bfloat16().contiguous()
    w = (torch.rand(D, device="cuda", generator=g) + 0.5).float()
    out = torch.zeros(B, H, S_pad, 128, device="cuda", dtype=torch.bfloat16)
    max2 = torch.zeros(B * H, device="cuda", dtype=torch.int32)
    _lib.check(lib.mi355_op_norm_rope_full(_stream(), _ptr(src), D, 0, _ptr(w), None, _ptr(out), M, H, S, S_pad, 1e-6, 0.7, _ptr(max2)),
               "op_norm_rope_full")
    torch.cuda.synchronize()
    got = max2.view(torch.float32).view(B, H).cpu()
    ref = out[:, :, :S].float().pow(2).sum(-1).amax(-1).cpu()
    assert torch.allclose(got, ref, rtol=1e-5, atol=0), (got, ref)
    x = src.float()
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w * 0.7
    assert ((out[:, :, :S].float().cpu() - y.view(B, S, H, 128).permute(0, 2, 1, 3).cpu()).abs().max() < 0.05)


def test_wan_self_attention_splits_static_and_running_max_by_the_measured_norms(wn):
    """mi355_tune_set(24, .): the kernel that stores q / k measures their largest row norm per (batch, head); (batch, head) pairs whose
    |q| |k| stays <= 60 run the static-softmax kernel, the others the running-max kernel -- inside ONE launch pair.  Here head 1's q / k norm
    weights are 4 x larger (its |q| |k| bound is in the hundreds: exp2 would overflow without a running max) while head 0 stays ordinary: the
    forward must match the oracle, and the all-running-max forward (key 24 = 0), to bf16 accuracy."""
    from mi355_flow import _lib
    from oracle import wan_ref as R
    lib = _lib.load()
    cfg_o = R.tiny_config()
    sd, cfg = _setup(wn, cfg_o, seed=77)
    H = cfg_o.num_attention_heads
    assert H >= 2
    for k_ in list(sd):
        if k_.endswith("attn1.norm_q.weight") or k_.endswith("attn1.norm_k.weight"):
            w = sd[k_].clone()
            w[128:256] *= 4.0
            sd[k_] = _bf(w)
    B, T, h, w_, Nt = 2, 2, 8, 12, 9
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 16, T, h, w_, generator=g).half()
    pe = _bf(torch.randn(B, Nt, cfg_o.text_dim, generator=g))
    t = torch.tensor([500.0])
    ref = R.wan_forward(sd, cfg_o, x.float(), t.expand(B), pe)
    outs = {}
    try:
        for key in (1, 0):
            _lib.check(lib.mi355_tune_set(24, key), "tune_set")
            eng = wn.WanEngine(cfg)
            eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
            eng.ready()
            outs[key] = eng.plan(B, 1, T, h, w_, Nt, 1).transformer_forward(x.cuda(), t, pe.cuda()).float().cpu()
            eng.close()
            assert torch.isfinite(outs[key]).all()
            rel = ((outs[key] - ref).norm() / ref.norm()).item()
            assert rel < 4e-2, (key, rel)            # (peaked softmaxes: the bf16 rounding of q / k shows more)
    finally:
        _lib.check(lib.mi355_tune_set(24, 1), "tune_set")
    assert ((outs[1] - outs[0]).norm() / outs[0].norm()).item() < 1e-2


@pytest.mark.parametrize("guidance", [1.0, 5.0])
def test_wan_rollout_matches_oracle_and_replays(wn, guidance):
    from oracle import wan_ref as R
    cfg_o = R.tiny_config()
    sd, cfg = _setup(wn, cfg_o, seed=12)
    sched = wn.UniPCMultistepSDEScheduler(flow_shift=3.0, noise_level=0.7, sde_steps=[0, 1, 2, 3], num_sde_steps=2, seed=42,
                                          dynamics_type="Flow-SDE")
    ad = wn.Wan2T2VNativeAdapter({k: v.cuda() for k, v in sd.items()}, cfg, sched, latent_storage_dtype="fp16")
    ad.rollout()
    B, Nt, N, H, W, frames = 2, 12, 5, 64, 96, 9           # latent grid (T, h, w) = (3, 8, 12)
    g = torch.Generator().manual_seed(4)
    pe = torch.randn(B, Nt, cfg_o.text_dim, generator=g).bfloat16()
    ne = torch.randn(B, Nt, cfg_o.text_dim, generator=g).bfloat16()
    cfg_on = guidance > 1.0
    torch.cuda.manual_seed(31)
    samples = ad.inference(prompt=["a", "b"], height=H, width=W, num_frames=frames, num_inference_steps=N, guidance_scale=guidance,
                           prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda() if cfg_on else None, compute_log_prob=True,
                           trajectory_indices="all")
    torch.cuda.manual_seed(31)
    T, h, w = 3, 8, 12
    init = torch.randn((B, 16, T, h, w), device="cuda", dtype=torch.float32).cpu()
    noise = torch.stack([torch.randn((B, 16, T, h, w), device="cuda", dtype=torch.float32) for _ in range(N)]).cpu()
    ts, sig = R.unipc_flow_schedule(N, 3.0)
    assert torch.equal(samples[0].timesteps.cpu(), ts)
    nl = ad.scheduler.host_noise_levels()
    assert sum(e > 0 for e in nl) == 2
    ref = R.rollout(sd, cfg_o, pe, ne if cfg_on else None, guidance, init, noise, ts, sig, nl, torch.float16)
    sde = [i for i in range(N) if nl[i] > 0]
    assert samples[0].all_latents.shape == (N + 1, 16, T, h, w) and samples[0].all_latents.dtype == torch.float16
    for b in range(B):
        got = samples[b].all_latents.float().cpu()
        for pos in range(N + 1):
            r = ref["all_latents"][pos, b].float()
            assert ((got[pos] - r).norm() / r.norm()).item() < 2e-2
        torch.testing.assert_close(samples[b].log_probs.cpu(), torch.stack([ref["log_probs"][i, b] for i in sde]), rtol=1e-3, atol=1e-4)
    # replay of a stored transition: ratio == 1 exactly
    i = sde[0]
    x_i = torch.stack([s.all_latents[i] for s in samples]).cuda()
    x_n = torch.stack([s.all_latents[i + 1] for s in samples]).cuda()
    out = ad.forward(t=samples[0].timesteps[i].reshape(1).expand(B).cuda(), latents=x_i, prompt_embeds=pe.cuda(),
                     negative_prompt_embeds=ne.cuda() if cfg_on else None, guidance_scale=guidance,
                     t_next=samples[0].timesteps[i + 1].reshape(1).expand(B).cuda(), next_latents=x_n, noise_level=nl[i],
                     compute_log_prob=True, return_kwargs=["log_prob"])
    old = torch.stack([s.log_probs[0] for s in samples]).cuda()
    assert torch.equal(torch.exp(out.log_prob - old), torch.ones_like(old))
    ad.engine.close()


@pytest.mark.parametrize("guidance,storage,N", [(5.0, "fp16", 6), (1.0, "bf16", 4), (4.0, None, 2)])
def test_wan_evaluation_mode_sampling_matches_oracle(wn, guidance, storage, N):
    """Evaluation-mode `inference()` (round 5: native -- per-step engine forwards + the UniPC multistep predictor-corrector as
    mi355_unipc_convert / mi355_op_lincomb with host-side coefficients, mi355_flow/unipc.py; reference wan2_t2v.py:346-375 over
    scheduler/unipc_multistep.py:282-285) against the oracle's evaluation loop (oracle/wan_ref.rollout_eval over oracle/unipc_ref.py, the
    published solver tensor by tensor; parity with diffusers itself unpinned).  Deterministic: two runs from one seed agree bit for bit and
    draw nothing but the initial latents."""
    from oracle import wan_ref as R
    cfg_o = R.tiny_config()
    sd, cfg = _setup(wn, cfg_o, seed=21)
    sched = wn.UniPCMultistepSDEScheduler(flow_shift=3.0, noise_level=0.7, sde_steps=[0, 1, 2], num_sde_steps=2, seed=42, dynamics_type="Flow-SDE")
    ad = wn.Wan2T2VNativeAdapter({k: v.cuda() for k, v in sd.items()}, cfg, sched, latent_storage_dtype=storage)
    ad.eval()
    assert ad.scheduler.is_eval
    B, Nt, H, W, frames = 2, 12, 64, 96, 9           # latent grid (T, h, w) = (3, 8, 12)
    g = torch.Generator().manual_seed(4)
    pe = torch.randn(B, Nt, cfg_o.text_dim, generator=g).bfloat16()
    ne = torch.randn(B, Nt, cfg_o.text_dim, generator=g).bfloat16()
    cfg_on = guidance > 1.0
    kw = dict(prompt=["a", "b"], height=H, width=W, num_frames=frames, num_inference_steps=N, guidance_scale=guidance, prompt_embeds=pe.cuda(),
              negative_prompt_embeds=ne.cuda() if cfg_on else None, compute_log_prob=False, trajectory_indices="all")
    torch.cuda.manual_seed(31)
    samples = ad.inference(**kw)
    after = torch.randn(4, device="cuda")
    torch.cuda.manual_seed(31)
    T, h, w = 3, 8, 12
    init = torch.randn((B, 16, T, h, w), device="cuda", dtype=torch.float32)
    assert torch.equal(after, torch.randn(4, device="cuda"))               # inference() drew the initial latents and nothing else
    torch.cuda.manual_seed(31)
    again = ad.inference(**kw)
    ts, sig = R.unipc_flow_schedule(N, 3.0)
    assert torch.equal(samples[0].timesteps.cpu(), ts)
    sdt = {"fp16": torch.float16, "bf16": torch.bfloat16, None: torch.float32}[storage]
    ref = R.rollout_eval(sd, cfg_o, pe, ne if cfg_on else None, guidance, init.cpu(), ts, sig, sdt)
    assert samples[0].all_latents.shape == (N + 1, 16, T, h, w) and samples[0].all_latents.dtype == sdt and samples[0].log_probs is None
    worst = 0.0
    for b in range(B):
        got = samples[b].all_latents.float().cpu()
        assert torch.equal(samples[b].all_latents, again[b].all_latents)
        assert torch.equal(got[0], ref["all_latents"][0, b].float())
        for pos in range(1, N + 1):
            r = ref["all_latents"][pos, b].float()
            worst = max(worst, ((got[pos] - r).norm() / r.norm()).item())
    print(f"Wan evaluation-mode sampling (guidance {guidance}, storage {storage}, {N} steps): worst latent rel-L2 vs the oracle loop {worst:.3e}")
    assert worst < 2e-2
    with pytest.raises(NotImplementedError, match="latents only"):
        ad.inference(**dict(kw, compute_log_prob=True))
    ad.engine.close()


def test_unipc_kernels_match_their_torch_statements(wn):
    """mi355_unipc_convert / mi355_op_lincomb against the torch statements of what they compute (the ones the CPU control-flow test runs)."""
    from mi355_flow import unipc as U
    from oracle import rollout_ref as RR
    g = torch.Generator().manual_seed(9)
    n = (3, 16, 2, 6, 10)
    vt, vu = torch.randn(n, generator=g).bfloat16(), torch.randn(n, generator=g).bfloat16()
    for sdt in (torch.float16, torch.bfloat16, torch.float32):
        x = (2 * torch.randn(n, generator=g)).to(sdt)
        for cfg_on in (False, True):
            got = U.unipc_convert(vt.cuda(), vu.cuda() if cfg_on else None, 4.5, x.cuda(), 0.8125).cpu()
            v = RR.cfg_combine_bf16(vu, vt, 4.5) if cfg_on else vt
            want = x.float() - (torch.tensor(0.8125) * v.float()).to(torch.bfloat16).float()
            assert got.dtype == torch.float32 and torch.equal(got, want), (sdt, cfg_on)
        # stored x0-predictions: fp32 beside an fp16 / fp32 sample, bf16 beside a bf16 sample (torch's promoted dtype of sample and prediction);
        # partial sums are rounded to the promoted dtype of the terms so far (bf16 + bf16 stays bf16; anything mixed is fp32)
        for mdt in (torch.float32, sdt):
            ms = [torch.randn(n, generator=g).to(mdt) for _ in range(3)]
            for k in range(1, 5):
                ts_, cs = [x] + ms[:k - 1], [0.91, -0.37, 0.52, -0.11][:k]
                got = U.lincomb([t.cuda() for t in ts_], cs, sdt).cpu()
                acc, prom = None, None
                for t, c in zip(ts_, cs):
                    term = (torch.tensor(c, dtype=torch.float32) * t.float()).to(t.dtype).float()
                    prom = t.dtype if prom is None else (prom if prom == t.dtype else torch.float32)
                    acc = term if acc is None else (acc + term).to(prom).float()
                assert got.dtype == sdt and torch.equal(got, acc.to(sdt)), (sdt, mdt, k)


def test_wan_errors(wn):
    with pytest.raises(RuntimeError, match="head_dim must be 128"):
        wn.WanEngine(wn.WanConfig(attention_head_dim=64))
    eng = wn.WanEngine(wn.WanConfig(num_layers=1, num_attention_heads=1, ffn_dim=128, text_dim=64))
    with pytest.raises(RuntimeError, match="must be even"):
        eng.plan(1, 1, 2, 5, 4, 8, 1)
    plan = eng.plan(1, 1, 1, 4, 4, 8, 1)
    with pytest.raises(RuntimeError, match="has not been bound"):
        plan.transformer_forward(torch.zeros(1, 16, 1, 4, 4).cuda(), torch.tensor([500.0]), torch.zeros(1, 8, 64).cuda())
    eng.close()


def test_wan_stepwise_callbacks_equal_fused_rollout(wn):
    from oracle import wan_ref as R
    cfg_o = R.tiny_config()
    sd, cfg = _setup(wn, cfg_o, seed=8)
    sched = wn.UniPCMultistepSDEScheduler(flow_shift=3.0, noise_level=0.7, sde_steps=[0, 1, 2], num_sde_steps=2, seed=1)
    ad = wn.Wan2T2VNativeAdapter({k: v.cuda() for k, v in sd.items()}, cfg, sched)
    ad.rollout()
    g = torch.Generator().manual_seed(1)
    pe = torch.randn(2, 9, cfg_o.text_dim, generator=g).bfloat16().cuda()
    ne = torch.randn(2, 9, cfg_o.text_dim, generator=g).bfloat16().cuda()
    kw = dict(prompt=["a", "b"], height=32, width=48, num_frames=5, num_inference_steps=4, guidance_scale=4.0, prompt_embeds=pe,
              negative_prompt_embeds=ne, compute_log_prob=True)
    torch.cuda.manual_seed(5)
    a = ad.inference(**kw)
    torch.cuda.manual_seed(5)
    b = ad.inference(**kw, extra_call_back_kwargs=["noise_pred"])
    for sa, sb in zip(a, b):
        assert torch.equal(sa.all_latents, sb.all_latents) and torch.equal(sa.log_probs, sb.log_probs)
        assert sb.extra_kwargs["noise_pred"].shape == (4,) + tuple(sa.all_latents.shape[1:])
    ad.engine.close()


def test_wan_full_width_blocks_at_4608_tokens(wn):
    """Wan2.1-T2V-1.3B WIDTH (D = 1536, 12 heads x 128, ffn 8960, T5 width 4096) with 2 blocks on a 4 x 48 x 96 latent grid
    (4 * 24 * 48 = 4608 video tokens: large-grid GEMMs, 8-wave attention128 with S_kv != S for the cross-attention), B = 1,
    CFG on (forward batch 2), vs the fp32 oracle (model body unpinned, oracle/wan_ref.py)."""
    from oracle import wan_ref as R
    cfg_o = R.WanConfig(num_layers=2)
    sd, cfg = _setup(wn, cfg_o, seed=77)
    eng = wn.WanEngine(cfg)
    eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    eng.ready()
    B, T, h, w, Nt = 1, 4, 48, 96, 226
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 16, T, h, w, generator=g).half()
    pe = _bf(torch.randn(B, Nt, cfg_o.text_dim, generator=g))
    ne = _bf(torch.randn(B, Nt, cfg_o.text_dim, generator=g))
    t = torch.tensor([601.0])
    got = eng.plan(B, 2, T, h, w, Nt, 1).transformer_forward(x.cuda(), t, ne.cuda(), pe.cuda())
    from _gpu_oracle import check_in_band
    pair = lambda sd_, x_, t_, ne_, pe_, quant=None: torch.cat([R.wan_forward(sd_, cfg_o, x_, t_, ne_, quant=quant), R.wan_forward(sd_, cfg_o, x_, t_, pe_, quant=quant)])  # noqa: E731
    check_in_band("Wan full-width 2 blocks, S = 4608, CFG pair", got, pair, sd, x.float(), t.expand(B), ne, pe)
    eng.close()


def test_wan_full_width_blocks_at_config_d_20280_tokens(wn):
    """BASELINE.json configs[3] at ITS OWN shape (reference wan2_t2v.py:426-543): 480 x 832 x 49 frames = a 13 x 60 x 104 latent grid =
    13 * 30 * 52 = 20 280 video tokens (not a multiple of 64: the last key tile is masked; operand offsets beyond 2^31 bytes in the
    q / k / V^T scatter at forward batch 2), Wan2.1-1.3B width, 2 blocks, CFG on, vs the fp32 oracle (model body unpinned)."""
    from oracle import wan_ref as R
    cfg_o = R.WanConfig(num_layers=2)
    sd, cfg = _setup(wn, cfg_o, seed=78)
    eng = wn.WanEngine(cfg)
    eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    eng.ready()
    B, T, h, w, Nt = 1, 13, 60, 104, 226
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, 16, T, h, w, generator=g).half()
    pe = _bf(torch.randn(B, Nt, cfg_o.text_dim, generator=g))
    ne = _bf(torch.randn(B, Nt, cfg_o.text_dim, generator=g))
    t = torch.tensor([601.0])
    got = eng.plan(B, 2, T, h, w, Nt, 1).transformer_forward(x.cuda(), t, ne.cuda(), pe.cuda())
    from _gpu_oracle import check_in_band, rel as rel_
    pair = lambda sd_, x_, t_, ne_, pe_, quant=None: torch.cat([R.wan_forward(sd_, cfg_o, x_, t_, ne_, quant=quant), R.wan_forward(sd_, cfg_o, x_, t_, pe_, quant=quant)])  # noqa: E731
    ref, _, r, band = check_in_band("Wan full-width 2 blocks, S = 20280 (config D), CFG pair", got, pair, sd, x.float(), t.expand(B), ne, pe)
    # per-frame: an indexing slip that only hits the far end of the 20 280-token axis must not hide in the global norm
    per_frame = [rel_(got[:, :, f], ref[:, :, f]) for f in range(T)]
    print(f"  worst frame {max(per_frame):.3e}")
    assert max(per_frame) < 1.5 * (1.5 * band + 1e-3), per_frame
    eng.close()


def test_wan22_two_expert_rollout_matches_oracle_and_replays(wn):
    """Wan2.2 two-expert pipelines (reference wan2_t2v.py:476-487): high-noise expert + `guidance_scale` while t >= boundary_ratio * 1000,
    low-noise expert + `guidance_scale_2` below (here <= 1: that expert runs without CFG).  Latents / log-probs vs the oracle's
    two-expert loop; replay of a step on EITHER side of the boundary reproduces the rollout log-prob bit for bit."""
    from oracle import wan_ref as R
    cfg_o = R.tiny_config()
    sd_hi, cfg = _setup(wn, cfg_o, seed=12)
    sd_lo, _ = _setup(wn, cfg_o, seed=99)
    sched = wn.UniPCMultistepSDEScheduler(flow_shift=3.0, noise_level=0.7, sde_steps=[0, 1, 2, 3, 4], num_sde_steps=5, seed=42, dynamics_type="Flow-SDE")
    ratio = 0.9                                   # boundary timestep 900: with N = 5, shift 3 the schedule crosses it after 2 steps
    ad = wn.Wan2T2VNativeAdapter({k: v.cuda() for k, v in sd_hi.items()}, cfg, sched, latent_storage_dtype="fp16",
                                 state_dict_2={k: v.cuda() for k, v in sd_lo.items()}, boundary_ratio=ratio)
    ad.rollout()
    B, Nt, N, H, W, frames = 2, 12, 5, 64, 96, 9
    g = torch.Generator().manual_seed(4)
    pe = torch.randn(B, Nt, cfg_o.text_dim, generator=g).bfloat16()
    ne = torch.randn(B, Nt, cfg_o.text_dim, generator=g).bfloat16()
    g1, g2 = 4.0, 1.0
    torch.cuda.manual_seed(31)
    samples = ad.inference(prompt=["a", "b"], height=H, width=W, num_frames=frames, num_inference_steps=N, guidance_scale=g1, guidance_scale_2=g2,
                           prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(), compute_log_prob=True, trajectory_indices="all")
    torch.cuda.manual_seed(31)
    T, h, w = 3, 8, 12
    init = torch.randn((B, 16, T, h, w), device="cuda", dtype=torch.float32).cpu()
    noise = torch.stack([torch.randn((B, 16, T, h, w), device="cuda", dtype=torch.float32) for _ in range(N)]).cpu()
    ts, sig = R.unipc_flow_schedule(N, 3.0)
    hi_steps = [i for i in range(N) if float(ts[i]) >= 900.0]
    assert 0 < len(hi_steps) < N, ts                                       # both experts are exercised
    nl = ad.scheduler.host_noise_levels()
    ref = R.rollout_two_expert(sd_hi, sd_lo, cfg_o, 900.0, pe, ne, g1, g2, init, noise, ts, sig, nl, torch.float16)
    for b in range(B):
        got = samples[b].all_latents.float().cpu()
        for pos in range(N + 1):
            r = ref["all_latents"][pos, b].float()
            assert ((got[pos] - r).norm() / r.norm()).item() < 2e-2, (b, pos)
        sde = [i for i in range(N) if nl[i] > 0]
        torch.testing.assert_close(samples[b].log_probs.cpu(), torch.stack([ref["log_probs"][i, b] for i in sde]), rtol=1e-3, atol=1e-4)
    # a single-expert run with the high-noise weights differs after the boundary: the second expert really ran
    one = wn.Wan2T2VNativeAdapter({k: v.cuda() for k, v in sd_hi.items()}, cfg, sched, latent_storage_dtype="fp16")
    one.rollout()
    torch.cuda.manual_seed(31)
    s1 = one.inference(prompt=["a", "b"], height=H, width=W, num_frames=frames, num_inference_steps=N, guidance_scale=g1, prompt_embeds=pe.cuda(),
                       negative_prompt_embeds=ne.cuda(), compute_log_prob=True, trajectory_indices="all")
    k = len(hi_steps)
    assert torch.equal(s1[0].all_latents[:k + 1], samples[0].all_latents[:k + 1]) and not torch.equal(s1[0].all_latents[k + 1], samples[0].all_latents[k + 1])
    # replay on both sides of the boundary: ratio == 1 exactly
    for i in (hi_steps[-1], hi_steps[-1] + 1):
        x_i = torch.stack([s.all_latents[i] for s in samples]).cuda()
        x_n = torch.stack([s.all_latents[i + 1] for s in samples]).cuda()
        out = ad.forward(t=samples[0].timesteps[i].reshape(1).expand(B).cuda(), latents=x_i, prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(),
                         guidance_scale=g1, guidance_scale_2=g2, t_next=(samples[0].timesteps[i + 1] if i + 1 < N else torch.zeros(())).reshape(1).expand(B).cuda(),
                         next_latents=x_n, noise_level=nl[i], compute_log_prob=True, return_kwargs=["log_prob"])
        old = torch.stack([s.log_probs[i] for s in samples]).cuda()
        assert torch.equal(out.log_prob, old), i
    ad.engine.close(); ad.engine_2.close(); one.engine.close()
