"""CPU self-consistency of the video-VAE oracle (oracle/wan_vae_ref.py): the whole-sequence formulation the engine implements equals the
published frame-by-frame algorithm with its feature cache (AutoencoderKLWan._decode), including the first-frame rule of the temporal
upsampler; one latent frame (Qwen-Image) reduces every causal 3x3x3 conv to its last temporal slice."""
import torch
import torch.nn.functional as F

from oracle import wan_vae_ref as V


def test_full_sequence_form_equals_the_chunked_feature_cache_algorithm():
    cfg = V.tiny_config()
    sd = V.make_synthetic_state_dict(cfg, seed=4)
    g = torch.Generator().manual_seed(0)
    for T, h, w in ((1, 4, 6), (2, 4, 4), (4, 2, 4)):
        z = torch.randn(1, 16, T, h, w, generator=g)
        with torch.no_grad():
            a, b = V.decode_full(sd, cfg, z), V.decode_chunked(sd, cfg, z)
        assert a.shape == b.shape == (1, 3, 1 + 4 * (T - 1), 8 * h, 8 * w)
        assert float((a - b).abs().max()) < 2e-5, (T, float((a - b).abs().max()))


def test_single_frame_uses_only_the_last_temporal_slice():
    cfg = V.tiny_config()
    sd = V.make_synthetic_state_dict(cfg, seed=5)
    z = torch.randn(2, 16, 1, 4, 4, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = V.decode_full(sd, cfg, z)
        sd2 = {k: v.clone() for k, v in sd.items()}
        for k, v in sd2.items():
            if k.endswith(".weight") and v.dim() == 5 and v.shape[2] == 3:
                v[:, :, :2] = 123.0                      # taps that only ever see the zero frames in front
        assert torch.equal(V.decode_full(sd2, cfg, z), ref)
        # ... i.e. a plain 2-D convolution with weight[:, :, 2]
        x = torch.randn(1, 16, 1, 5, 5)
        wt, b = sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"]
        assert torch.allclose(V.causal_conv3d(x, wt, b)[:, :, 0], F.conv2d(x[:, :, 0], wt[:, :, 2], b, padding=1), atol=1e-5)


def test_shapes_and_flops_of_the_wan21_geometry():
    cfg = V.WAN21
    assert cfg.dims == [384, 384, 384, 192, 96]
    assert cfg.up_plan() == [(384, 384, "upsample3d"), (192, 384, "upsample3d"), (192, 192, "upsample2d"), (96, 96, None)]
    sh = V.state_dict_shapes(cfg)
    assert sh["decoder.up_blocks.1.resnets.0.conv_shortcut.weight"] == (384, 192, 1, 1, 1)
    assert sh["decoder.up_blocks.0.upsamplers.0.time_conv.weight"] == (768, 384, 3, 1, 1)
    assert sh["decoder.up_blocks.2.upsamplers.0.resample.1.weight"] == (96, 192, 3, 3)
    assert "decoder.up_blocks.2.upsamplers.0.time_conv.weight" not in sh and "decoder.up_blocks.3.upsamplers.0.resample.1.weight" not in sh
    n = sum(int(torch.tensor(s).prod()) for s in sh.values())
    assert 50e6 < n < 80e6                                # ~ 73 M decoder parameters
    assert V.decode_flops(cfg, 13, 60, 104) > 1e14        # 480 x 832 x 49 frames: > 100 TFLOP per clip
