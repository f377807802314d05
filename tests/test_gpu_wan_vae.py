"""GPU parity of the causal 3-D video VAE decode (Wan `decode_latents`, wan2_t2v.py:215-230; Qwen-Image
`decode_latents`, qwen_image.py:197-213) against the CPU oracle (oracle/wan_vae_ref.py; decoder body unpinned, see its header) and plain
torch references of the generalised convolution.  Everything goes through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.bfloat16().float()


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def vm():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mi355_flow import vae
    return vae


@pytest.mark.parametrize("B,T,H,W,Ci,Co,kt,ks,up,skip", [
    (1, 3, 6, 5, 64, 64, 3, 3, 0, 0),        # causal 3x3x3
    (2, 4, 8, 8, 128, 96, 3, 3, 0, 0),       # two samples: the causal padding restarts per sample; ragged N
    (1, 5, 4, 6, 64, 128, 3, 1, 0, 1),       # time_conv over frames 1.. as their own sequence
    (2, 3, 8, 4, 64, 64, 1, 3, 1, 0),        # per-frame 3x3 with the nearest-2x upsample folded in
    (1, 1, 16, 16, 192, 64, 1, 3, 0, 0),     # single frame, K = 9 * 192
    (1, 9, 16, 24, 64, 64, 3, 3, 0, 0),      # > 1 tile of rows, frames straddle tiles
])
def test_causal_conv_matches_torch(vm, B, T, H, W, Ci, Co, kt, ks, up, skip):
    g = torch.Generator().manual_seed(T * 100 + H + Ci)
    Hin, Win = (H // 2, W // 2) if up else (H, W)
    T_in = T + skip
    x = _bf(torch.randn(B, T_in, Hin, Win, Ci, generator=g))
    w = _bf(torch.randn(Co, Ci, kt, ks, ks, generator=g) * (1.0 / (Ci * kt * ks * ks)) ** 0.5)
    b = torch.randn(Co, generator=g) * 0.1
    wp = vm.op_conv_repack(w.cuda(), Ci)
    assert tuple(wp.shape) == (Co, kt * ks * ks, Ci)
    got = vm.op_conv3d_causal(x.bfloat16().cuda(), wp, b.cuda(), kt, ks, frames=T, upsample=bool(up), skip_frames=skip).float().cpu()
    xs = x[:, skip:].permute(0, 4, 1, 2, 3)                                   # (B, C, T, H, W), the sub-sequence
    if up:
        xs = F.interpolate(xs.reshape(B, Ci * T, Hin, Win), scale_factor=2.0, mode="nearest").reshape(B, Ci, T, H, W)
    xs = F.pad(xs, (ks // 2, ks // 2, ks // 2, ks // 2, kt - 1, 0))
    ref = F.conv3d(xs, w, b).permute(0, 2, 3, 4, 1)
    assert got.shape == ref.shape
    assert _rel(got, ref) < 5e-3, _rel(got, ref)
    # residual epilogue
    r = _bf(torch.randn(ref.shape, generator=g))
    got2 = vm.op_conv3d_causal(x.bfloat16().cuda(), wp, b.cuda(), kt, ks, frames=T, residual=r.bfloat16().cuda(), upsample=bool(up),
                               skip_frames=skip).float().cpu()
    assert _rel(got2, ref + r) < 5e-3


def test_wan_rms_matches_oracle(vm):
    from oracle import wan_vae_ref as V
    g = torch.Generator().manual_seed(3)
    for C, Cp in ((96, 128), (384, 384), (32, 64)):
        x = torch.zeros(50, Cp)
        x[:, :C] = _bf(torch.randn(50, C, generator=g) * 3)
        gamma = torch.zeros(Cp)
        gamma[:C] = 1 + 0.2 * torch.randn(C, generator=g)
        for silu in (False, True):
            got = vm.op_wan_rms(x.bfloat16().cuda(), gamma.cuda(), C, silu).float().cpu()
            ref = V.rms_norm(x[:, :C, None, None, None], gamma[:C].view(C, 1, 1, 1))[:, :, 0, 0, 0]
            ref = F.silu(ref) if silu else ref
            assert (got[:, :C] - ref).abs().max().item() < 3e-2 and _rel(got[:, :C], ref) < 4e-3
            assert float(got[:, C:].abs().max()) == 0.0 if Cp > C else True


def _decoder(vm, cfg_o, seed):
    from oracle import wan_vae_ref as V
    sd = {k: _bf(v) for k, v in V.make_synthetic_state_dict(cfg_o, seed=seed).items()}
    cfg = vm.WanVAEConfig(base_dim=cfg_o.base_dim, dim_mult=tuple(cfg_o.dim_mult), num_res_blocks=cfg_o.num_res_blocks,
                          temperal_downsample=tuple(cfg_o.temperal_downsample))
    dec = vm.WanVAEDecoder(cfg)
    # a full VAE state dict carries encoder / quant_conv keys as well: they are ignored, decoder names must all be present
    extra = {"encoder.conv_in.weight": torch.zeros(4, 3, 3, 3, 3), "quant_conv.weight": torch.zeros(32, 32, 1, 1, 1)}
    dec.bind_state_dict({**{k: v.cuda() for k, v in sd.items()}, **extra})
    dec.ready()
    return sd, dec


@pytest.mark.parametrize("T,h,w,B", [(1, 8, 8, 2), (3, 4, 8, 1), (2, 8, 4, 2)])
def test_tiny_decode_matches_oracle(vm, T, h, w, B):
    """Tiny widths (64 / 64 / 64 / 32): one latent frame (the Qwen-Image case: 2-D path on the last temporal slices) and short clips
    (frame-0 rule of the temporal upsampler, causal padding per sample), raw and post-processed outputs."""
    from oracle import wan_vae_ref as V
    cfg_o = V.tiny_config()
    sd, dec = _decoder(vm, cfg_o, seed=4 + T)
    g = torch.Generator().manual_seed(10 * T + h)
    lat = torch.randn(B, 16, T, h, w, generator=g).half()
    with torch.no_grad():
        ref_raw = V.decode_latents(sd, cfg_o, lat, postprocess=False)            # (B, 3, F, H, W) in [-1, 1]
        ref_pp = V.decode_latents(sd, cfg_o, lat, postprocess=True)              # (B, F, 3, H, W) in [0, 1]
    Fr = 1 + 4 * (T - 1)
    raw = dec.decode(lat.cuda(), postprocess=False, out_dtype=torch.float32, max_batch=B)
    assert tuple(raw.shape) == (B, Fr, 3, 8 * h, 8 * w) and raw.dtype == torch.float32
    r = _rel(raw.permute(0, 2, 1, 3, 4), ref_raw)
    assert r < 3e-2, r
    assert float(raw.abs().max()) <= 1.0
    pp = dec.decode(lat.cuda(), postprocess=True, out_dtype=torch.bfloat16, max_batch=1)        # one sample per launch sequence
    assert pp.dtype == torch.bfloat16 and float(pp.min()) >= 0.0 and float(pp.max()) <= 1.0
    assert (pp.float().cpu() - ref_pp).abs().mean().item() < 6e-3
    # already de-normalised latents (what `vae.decode` itself takes)
    z = V.denormalise(lat, cfg_o)
    raw2 = dec.decode(z.cuda(), postprocess=False, out_dtype=torch.float32, denormalise=False)
    assert _rel(raw2, raw) < 1e-2
    dec.close()


def test_wan21_width_decode_matches_oracle(vm):
    """The released geometry (96 / 192 / 384 / 384, 73 M decoder parameters; channel widths 96 and 192 exercise the padded layouts) on a
    short low-resolution clip: 2 latent frames of 8 x 16 -> 5 frames of 64 x 128."""
    from oracle import wan_vae_ref as V
    sd, dec = _decoder(vm, V.WAN21, seed=2)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 16, 2, 8, 16, generator=g).half()
    with torch.no_grad():
        ref, stages = V.decode_full(sd, V.WAN21, V.denormalise(lat, V.WAN21), return_stages=True)
    got = dec.decode(lat.cuda(), postprocess=False, out_dtype=torch.float32).permute(0, 2, 1, 3, 4)
    r = _rel(got, ref)
    print(f"Wan2.1 VAE width, 2 latent frames 8x16 -> 5 x 64x128: rel-L2 {r:.3e} (|ref| mean {float(ref.abs().mean()):.3f})")
    assert tuple(got.shape) == (1, 3, 5, 64, 128) and r < 3e-2, r
    # single frame (Qwen-Image): (B, 3, H, W) through the packed-latent helper
    from mi355_flow.flux import pack_latents
    from mi355_flow.qwen import decode_packed_latents
    lat1 = torch.randn(2, 16, 16, 16, generator=g).bfloat16()
    with torch.no_grad():
        ref1 = V.decode_latents(sd, V.WAN21, lat1.unsqueeze(2), postprocess=True)[:, 0]
    img = decode_packed_latents(dec, pack_latents(lat1).cuda(), 128, 128)
    assert tuple(img.shape) == (2, 3, 128, 128) and (img.float().cpu() - ref1).abs().mean().item() < 6e-3
    dec.close()


def test_video_vae_errors(vm):
    with pytest.raises(RuntimeError, match="multiple of 64"):
        vm.WanVAEDecoder(vm.WanVAEConfig(base_dim=24))
    from oracle import wan_vae_ref as V
    cfg_o = V.tiny_config()
    cfg = vm.WanVAEConfig(base_dim=32, dim_mult=(1, 2, 2, 2))
    dec = vm.WanVAEDecoder(cfg)
    sd = V.make_synthetic_state_dict(cfg_o)
    with pytest.raises(KeyError):
        dec.bind_state_dict({k: v.cuda() for k, v in sd.items() if "time_conv" not in k})
    with pytest.raises(ValueError):
        dec.decode(torch.zeros(1, 16, 8, 8, device="cuda"))
    dec.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    with pytest.raises(RuntimeError, match="multiple of 8"):
        dec.decode(torch.zeros(1, 16, 1, 3, 5, device="cuda"))
    with pytest.raises(ValueError):
        vm.WanVAEConfig.from_hf({"is_residual": True})
    dec.close()
