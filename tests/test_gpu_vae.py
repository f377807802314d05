"""GPU parity of the native VAE decode against the CPU oracle (oracle/vae_ref.py) and plain
torch fp32 references of the individual ops.  Everything goes through the C ABI (libmi355flow.so)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.bfloat16().float()


@pytest.fixture(scope="module")
def vae_mod():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import vae
    return vae


@pytest.mark.parametrize("B,H,W,Cin,Cout,up,res", [
    (2, 16, 16, 64, 128, False, False),
    (1, 24, 20, 128, 64, False, True),       # non power-of-two image, N < tile
    (2, 16, 16, 128, 256, True, False),      # nearest-2x upsample folded into the gather
    (1, 10, 6, 64, 3, False, False),         # tiny N, ragged M
    (4, 128, 128, 128, 128, False, True),    # 256x128 tiles
    (2, 128, 128, 64, 256, False, False),    # 256x256 tiles
    (2, 256, 256, 128, 128, False, False),   # 512x128 tiles (the full-resolution layers)
])
def test_conv3x3_matches_torch(vae_mod, B, H, W, Cin, Cout, up, res):
    g = torch.Generator().manual_seed(B * 1000 + H + Cin + Cout)
    hin, win = (H // 2, W // 2) if up else (H, W)
    x = _bf(torch.randn(B, Cin, hin, win, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5)
    b = 0.1 * torch.randn(Cout, generator=g)
    r = _bf(torch.randn(B, Cout, H, W, generator=g)) if res else None
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, padding=1)
    if res:
        ref = _bf(ref) + r
    wp = vae_mod.op_conv_repack(w.cuda())
    assert wp.shape == (Cout, 9, Cin)
    rr = r.permute(0, 2, 3, 1).contiguous().bfloat16().cuda() if res else None
    out = vae_mod.op_conv3x3(x.permute(0, 2, 3, 1).contiguous().bfloat16().cuda(), wp, b.cuda(), rr, up)
    got = out.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    assert err.max().item() <= 2e-2 * max(1.0, ref.abs().max().item()), err.max().item()
    assert (err.mean() / ref.abs().mean()).item() < 3e-3


def test_conv3x3_residual_in_place(vae_mod):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 16, 64, generator=g).bfloat16().cuda()
    w = vae_mod.op_conv_repack((torch.randn(64, 64, 3, 3, generator=g) / 24).cuda())
    b = torch.zeros(64).cuda()
    r = torch.randn(1, 16, 16, 64, generator=g).bfloat16().cuda()
    ref = vae_mod.op_conv3x3(x, w, b, r.clone())
    from mi355_flow import _lib
    from mi355_flow.vae import _ptr, _stream
    buf = r.clone()
    _lib.check(_lib.load().mi355_op_conv3x3(_stream(), _ptr(x), _ptr(w), _ptr(b), _ptr(buf), _ptr(buf), 1, 16, 16, 64, 64, 0))
    assert torch.equal(buf, ref)


@pytest.mark.parametrize("B,HW,C,silu", [(2, 64, 64, True), (1, 4096, 128, True), (2, 1000, 512, False), (1, 65536, 256, True)])
def test_group_norm_matches_torch(vae_mod, B, HW, C, silu):
    g = torch.Generator().manual_seed(HW + C)
    x = _bf(torch.randn(B, HW, C, generator=g) * 2 + 0.5)
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.3 * torch.randn(C, generator=g)
    ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, 1e-6).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    out = vae_mod.op_group_norm(x.bfloat16().cuda(), gamma.cuda(), beta.cuda(), 32, 1e-6, silu).float().cpu()
    err = (out - ref).abs()
    assert err.max().item() <= 1e-2 * max(1.0, ref.abs().max().item())
    out2 = vae_mod.op_group_norm(x.bfloat16().cuda(), gamma.cuda(), beta.cuda(), 32, 1e-6, silu).float().cpu()
    assert torch.equal(out, out2)   # deterministic reduction order


def _decode_case(vae_mod, cfg_o, h, w, B, lat_dtype, seed):
    from oracle import vae_ref as V
    sd = V.make_synthetic_state_dict(cfg_o, seed)
    g = torch.Generator().manual_seed(seed + 1)
    lat = (torch.randn(B, cfg_o.latent_channels, h, w, generator=g) * 1.2).to(lat_dtype)
    ref_raw = V.vae_decode(sd, cfg_o, lat, quant=_bf, postprocess=False)
    ref_img = V.vae_decode(sd, cfg_o, lat, quant=_bf, postprocess=True)
    cfg = vae_mod.VAEConfig(cfg_o.latent_channels, cfg_o.out_channels, tuple(cfg_o.block_out_channels), cfg_o.layers_per_block,
                            cfg_o.norm_num_groups, cfg_o.eps, cfg_o.scaling_factor, cfg_o.shift_factor)
    dec = vae_mod.VAEDecoder(cfg)
    dec.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    dec.ready()
    raw = dec.decode(lat.cuda(), postprocess=False, out_dtype=torch.float32).cpu()
    img = dec.decode(lat.cuda(), postprocess=True, out_dtype=torch.bfloat16).float().cpu()
    dec.close()
    return raw, img, ref_raw, ref_img


@pytest.mark.parametrize("h,w,B,lat_dtype", [(8, 8, 2, torch.float16), (16, 8, 1, torch.float32), (8, 16, 3, torch.bfloat16)])
def test_decode_tiny_matches_oracle(vae_mod, h, w, B, lat_dtype):
    from oracle import vae_ref as V
    raw, img, ref_raw, ref_img = _decode_case(vae_mod, V.tiny_config(), h, w, B, lat_dtype, 7)
    assert raw.shape == ref_raw.shape == (B, 3, 8 * h, 8 * w)
    rel = ((raw - ref_raw).pow(2).mean().sqrt() / ref_raw.pow(2).mean().sqrt()).item()
    assert rel < 2e-2, rel                      # bf16 chain of ~20 layers: independent roundings diverge by ~1e-2
    assert (raw - ref_raw).abs().max().item() < 0.15 * max(1.0, ref_raw.abs().max().item())
    assert img.min().item() >= 0.0 and img.max().item() <= 1.0
    assert (img - ref_img).abs().mean().item() < 1e-2


def test_decode_sd3_vae_matches_oracle(vae_mod):
    """Full SD3 decoder geometry (49.5 M parameters) on 32x32 latents -> 256x256 images."""
    from oracle import vae_ref as V
    raw, img, ref_raw, ref_img = _decode_case(vae_mod, V.SD3_VAE, 32, 32, 2, torch.float16, 11)
    rel = ((raw - ref_raw).pow(2).mean().sqrt() / ref_raw.pow(2).mean().sqrt()).item()
    assert rel < 3e-2, rel
    assert (img - ref_img).abs().mean().item() < 1e-2


def test_decode_batch_chunking_and_determinism(vae_mod):
    from oracle import vae_ref as V
    cfg_o = V.tiny_config()
    sd = V.make_synthetic_state_dict(cfg_o, 3)
    dec = vae_mod.VAEDecoder(vae_mod.VAEConfig(block_out_channels=tuple(cfg_o.block_out_channels), layers_per_block=1))
    dec.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    lat = torch.randn(5, 16, 8, 8, generator=torch.Generator().manual_seed(1)).half().cuda()
    a = dec.decode(lat, max_batch=5)
    b = dec.decode(lat, max_batch=2)
    c = dec.decode(lat, max_batch=5)
    assert torch.equal(a, c)
    assert torch.equal(a, b)          # per-sample results do not depend on how the batch is chunked
    dec.close()


def test_decode_errors(vae_mod):
    dec = vae_mod.VAEDecoder(vae_mod.VAEConfig(block_out_channels=(64, 64), layers_per_block=1))
    lat = torch.zeros(1, 16, 8, 8).cuda()
    with pytest.raises(RuntimeError, match="has not been bound"):
        dec.decode(lat)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        dec.decode(torch.zeros(1, 16, 6, 6).cuda())
    with pytest.raises(ValueError):
        dec.decode(torch.zeros(1, 16, 8, 8))      # CPU tensor: no fallback
    dec.close()


def test_adapter_inference_attaches_native_images(vae_mod):
    """`inference()` ends with decode_latents(final latents, 'pt') (sd3_5.py:307): with a bound VAE the samples carry images."""
    from oracle import mmditx_ref as M, vae_ref
    from mi355_flow.adapter import SD3_5NativeAdapter
    from mi355_flow.engine import TransformerConfig
    cfg_o = M.tiny_config(num_layers=2, num_heads=2, dual_layers=(0,), joint_attention_dim=128, pooled_projection_dim=128,
                          pos_embed_max_size=24)
    sd = M.make_synthetic_state_dict(cfg_o, seed=5, std=0.05)
    tc = TransformerConfig(num_layers=2, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24,
                           dual_layers=(0,))
    vcfg_o = vae_ref.tiny_config()
    vsd = vae_ref.make_synthetic_state_dict(vcfg_o, 21)
    vcfg = vae_mod.VAEConfig(block_out_channels=tuple(vcfg_o.block_out_channels), layers_per_block=vcfg_o.layers_per_block)
    ad = SD3_5NativeAdapter({k: v.cuda() for k, v in sd.items()}, tc, vae_state_dict={k: v.cuda() for k, v in vsd.items()},
                            vae_config=vcfg)
    ad.rollout()
    g = torch.Generator().manual_seed(3)
    B = 2
    pe = torch.randn(B, 7, 128, generator=g).bfloat16().cuda()
    pp = torch.randn(B, 128, generator=g).bfloat16().cuda()
    samples = ad.inference(prompt=["a", "b"], height=64, width=64, num_inference_steps=3, guidance_scale=1.0,
                           prompt_embeds=pe, pooled_prompt_embeds=pp, trajectory_indices="all")
    assert len(samples) == B
    for s in samples:
        assert s.image is not None and tuple(s.image.shape) == (3, 64, 64) and s.image.dtype == torch.bfloat16
        assert 0.0 <= s.image.float().min().item() and s.image.float().max().item() <= 1.0
    final = torch.stack([s.all_latents[-1] for s in samples])
    ref = vae_ref.vae_decode(vsd, vcfg_o, final.float().cpu(), quant=_bf, postprocess=True)
    got = torch.stack([s.image for s in samples]).float().cpu()
    assert (got - ref).abs().mean().item() < 1e-2
    ad.engine.close()
