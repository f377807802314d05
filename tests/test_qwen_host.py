"""CPU tests of the Qwen-Image host logic (mi355_flow/qwen.py): prompt padding against the REFERENCE's own `_pad_batch_prompt`, the
`[negative | positive]` forward-batch assembly, the timestep the network receives, the scheduler mirror's terminal stretch, and the
video-VAE config plumbing.  No compute calls (no GPU)."""
import pytest
import torch

from mi355_flow import qwen as QW


def _ragged(lens, J=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(n, J, generator=g) for n in lens], [torch.ones(n, dtype=torch.long) for n in lens]


def test_pad_batch_prompt_equals_the_references():
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    from flow_factory.models.qwen_image.qwen_image import QwenImageAdapter
    fake_self = type("S", (), {"_standardize_data": QwenImageAdapter._standardize_data, "tokenizer": None})()
    fake_self._standardize_data = lambda *a, **k: QwenImageAdapter._standardize_data(fake_self, *a, **k)
    emb, mask = _ragged([5, 9, 3])
    lens_r, mask_r, emb_r, _ = QwenImageAdapter._pad_batch_prompt(fake_self, prompt_embeds_mask=mask, prompt_embeds=emb, device=torch.device("cpu"))
    lens, m, e = QW.pad_batch_prompt(mask, emb, torch.device("cpu"))
    assert lens == [int(v) for v in lens_r] and torch.equal(m, mask_r) and torch.equal(e, emb_r)
    # padded-batch input with trailing padding beyond the longest prompt: truncated to it
    pe = torch.zeros(3, 12, 8)
    pm = torch.zeros(3, 12, dtype=torch.long)
    for b, (x, k) in enumerate(zip(emb, mask)):
        pe[b, :x.shape[0]], pm[b, :k.shape[0]] = x, k
    lens_r, mask_r, emb_r, _ = QwenImageAdapter._pad_batch_prompt(fake_self, prompt_embeds_mask=pm, prompt_embeds=pe, device=torch.device("cpu"))
    lens, m, e = QW.pad_batch_prompt(pm, pe, torch.device("cpu"))
    assert lens == [int(v) for v in lens_r] == [5, 9, 3] and e.shape == (3, 9, 8) and torch.equal(m, mask_r) and torch.equal(e, emb_r)


def test_forward_text_orders_negative_first_and_pads_to_the_plan_length():
    mix = QW.QwenRolloutMixin()
    emb, mask = _ragged([5, 40, 3])
    nemb, nmask = _ragged([2, 2, 2], seed=1)
    n_cfg, embeds, lens, pos, neg = mix._forward_text(emb, mask, nemb, nmask, 4.0, torch.device("cpu"))
    assert n_cfg == 2 and embeds.shape == (6, 64, 8) and embeds.dtype == torch.bfloat16       # 40 -> TEXT_PAD multiple
    assert lens == [2, 2, 2, 5, 40, 3]
    assert torch.equal(embeds[3, :5].float(), emb[0].bfloat16().float()) and float(embeds[3, 5:].abs().max()) == 0.0
    assert torch.equal(embeds[0, :2].float(), nemb[0].bfloat16().float())
    # no CFG: guidance <= 1, or no negative prompt (qwen_image.py:499-507)
    for kw in (dict(g=1.0, ne=nemb, nm=nmask), dict(g=4.0, ne=None, nm=None), dict(g=4.0, ne=nemb, nm=None)):
        n_cfg, embeds, lens, _, neg = mix._forward_text(emb, mask, kw["ne"], kw["nm"], kw["g"], torch.device("cpu"))
        assert n_cfg == 1 and neg is None and lens == [5, 40, 3] and embeds.shape[0] == 3
    with pytest.raises(ValueError):
        mix._forward_text(emb, mask, nemb[:2], nmask[:2], 4.0, torch.device("cpu"))


def test_model_timestep_is_the_value_the_reference_network_embeds():
    """qwen_image.py:497 `timestep = t.to(latents.dtype)`, :534 `timestep / 1000` in that dtype; Timesteps(scale=1000) multiplies back in fp32."""
    from oracle import qwen_ref as Q
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        t = torch.tensor([1000.0, 937.5, 750.25, 20.0])
        tm = QW.model_timestep(t, dt)
        want = (t.to(dt) / 1000).float()
        a = Q.timestep_embedding(want, 256)                      # scale * (t * freqs)
        half = 128
        freqs = torch.exp(-torch.log(torch.tensor(10000.0)) * torch.arange(half, dtype=torch.float32) / half)
        b = torch.cat([(tm[:, None] * freqs[None]).cos(), (tm[:, None] * freqs[None]).sin()], dim=-1)   # what launch_time_proj computes
        assert float((a - b).abs().max()) < 2e-4


def test_scheduler_mirror_terminal_stretch_and_dynamic_shift():
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler, set_scheduler_timesteps
    s = FlowMatchEulerDiscreteSDEScheduler(shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9, max_image_seq_len=8192,
                                           shift_terminal=0.02)
    ts = set_scheduler_timesteps(s, 10, seq_len=4096)
    sig = s.sigmas
    assert len(ts) == 10 and abs(float(sig[0]) - 1.0) < 1e-6 and abs(float(sig[9]) - 0.02) < 1e-6 and float(sig[10]) == 0.0
    assert all(float(sig[i]) > float(sig[i + 1]) for i in range(10))
    s2 = FlowMatchEulerDiscreteSDEScheduler(shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9, max_image_seq_len=8192)
    set_scheduler_timesteps(s2, 10, seq_len=4096)
    assert float(s2.sigmas[9]) > 0.02 + 1e-3                         # without the stretch the last sigma is the shifted 1/N


def test_video_vae_config_plumbing():
    from mi355_flow.vae import WanVAEConfig
    c = WanVAEConfig.from_hf({"z_dim": 16, "base_dim": 96, "dim_mult": [1, 2, 4, 4], "num_res_blocks": 2, "temperal_downsample": [False, True, True],
                              "latents_mean": [0.5] * 16, "latents_std": [2.0] * 16, "attn_scales": []})
    cc = c.to_c()
    assert list(cc.temporal_upsample) == [1, 1, 0] and list(cc.dim_mult) == [1, 2, 4, 4] and cc.latents_std[15] == 2.0
    assert [c.num_frames(t) for t in (1, 2, 13, 21)] == [1, 5, 49, 81]
    with pytest.raises(ValueError):
        WanVAEConfig.from_hf({"is_residual": True})
    with pytest.raises(ValueError):
        WanVAEConfig(dim_mult=(1, 2, 4)).to_c()
