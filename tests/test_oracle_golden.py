"""Pins the oracle restatements against fixtures produced by the reference's OWN code
(tests/golden/*.npz, generator: oracle/make_golden.py)."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import advantage_ref, mmditx_ref, rollout_ref, scheduler_ref

DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def _cases():
    z = _load("scheduler_steps.npz")
    return z, int(z["num_cases"][0])


def test_scheduler_step_matches_reference():
    z, n = _cases()
    assert n >= 40
    seen = set()
    for ci in range(n):
        k = f"c{ci}"
        dyn, sd_name, i, eta, clp, t, t_next, smax = [str(x) for x in z[k + "_meta"]]
        eta, clp, t, t_next, smax = float(eta), bool(int(clp)), float(t), float(t_next), float(smax)
        seen.add((dyn, sd_name))
        lat = torch.from_numpy(z[k + "_latents"]).to(DT[sd_name])
        v = torch.from_numpy(z[k + "_noise_pred"]).to(torch.bfloat16)
        eps = torch.from_numpy(z[k + "_eps"])
        o = scheduler_ref.sde_step(v, lat, torch.tensor(t) / 1000, torch.tensor(t_next) / 1000, eta, dyn,
                                   sigma_max=smax, variance_noise=eps, compute_log_prob=clp)
        # bitwise: same torch ops in the same order
        assert np.array_equal(o["next_latents"].numpy(), z[k + "_next"]), (ci, dyn, sd_name)
        assert np.array_equal(o["next_latents_mean"].numpy(), z[k + "_mean"])
        assert np.array_equal(o["std_dev_t"].numpy(), z[k + "_std"])
        assert np.array_equal(o["dt"].numpy(), z[k + "_dt"])
        if clp:
            assert np.array_equal(o["log_prob"].numpy(), z[k + "_logp"])
        if (k + "_replay_logp") in z.files:
            nxt = o["next_latents"].to(DT[sd_name])
            o2 = scheduler_ref.sde_step(v, lat, torch.tensor(t) / 1000, torch.tensor(t_next) / 1000, eta, dyn,
                                        sigma_max=smax, next_latents=nxt)
            assert np.array_equal(o2["log_prob"].numpy(), z[k + "_replay_logp"])
            # train/inference consistency invariant: ratio == 1 (train_inference_consistency.md:20-29)
            assert np.array_equal(o2["log_prob"].numpy(), z[k + "_logp"])
    assert len(seen) == 12


def test_schedule_matches_reference():
    z = _load("schedule.npz")
    for N in (4, 10, 28):
        ts, sig = scheduler_ref.make_schedule(N, shift=3.0)
        assert np.array_equal(ts.numpy(), z[f"static_N{N}_timesteps"])
        assert np.array_equal(sig.numpy(), z[f"static_N{N}_sigmas"])
        for seq in (256, 1024, 4096):
            ts, sig = scheduler_ref.make_schedule(N, shift=1.0, use_dynamic_shifting=True, seq_len=seq)
            np.testing.assert_allclose(ts.numpy(), z[f"dyn_N{N}_S{seq}_timesteps"], rtol=1e-6)
            np.testing.assert_allclose(sig.numpy(), z[f"dyn_N{N}_S{seq}_sigmas"], rtol=1e-6)
    ts, _ = scheduler_ref.make_schedule(4, shift=3.0)
    assert ts.tolist() == [1000.0, 900.0, 750.0, 500.0]  # SURVEY.md 8(a) A4 probe


def test_sde_step_selection_matches_reference():
    z = _load("schedule.npz")
    cfgs = [ast.literal_eval(str(c)) for c in z["select_cfg"]]
    rows = [ast.literal_eval(str(r)) for r in z["select_rows"]]
    assert len(rows) == len(cfgs) * 24
    for ci, seed, cur, nl in rows:
        steps, n = cfgs[ci]
        got = scheduler_ref.current_sde_steps(steps, n, seed, 10)
        assert got.tolist() == cur
        np.testing.assert_allclose(scheduler_ref.noise_levels(10, got, 0.7).numpy(), np.array(nl, np.float32))


def test_advantages_match_reference():
    z = _load("advantages.npz")
    rewards = {"clip": z["clip"], "pick": z["pick"]}
    w = {"clip": 1.0, "pick": 0.5}
    gi = advantage_ref.group_indices_from_ids(z["ids"])
    K = int(z["K"][0])
    for world in (1, 2):
        np.testing.assert_allclose(advantage_ref.weighted_sum(rewards, w, gi, K, True), z[f"sum_gstd_w{world}"], rtol=3e-6, atol=3e-6)
        np.testing.assert_allclose(advantage_ref.weighted_sum(rewards, w, gi, K, False), z[f"sum_lstd_w{world}"], rtol=3e-6, atol=3e-6)
        np.testing.assert_allclose(advantage_ref.gdpo(rewards, w, gi), z[f"gdpo_w{world}"], rtol=3e-6, atol=3e-6)


def test_mmdit_oracle_self_consistency():
    """[SELF] fixture: guards the (unpinned) denoiser restatement against silent drift."""
    z = _load("mmdit_tiny_self.npz")
    cfg = mmditx_ref.tiny_config()
    sd = mmditx_ref.make_synthetic_state_dict(cfg, seed=1234, std=0.08)
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(2, 16, 8, 8, generator=g)
    e = torch.randn(2, 13, cfg.joint_attention_dim, generator=g)
    p = torch.randn(2, cfg.pooled_projection_dim, generator=g)
    y = mmditx_ref.mmdit_forward(sd, cfg, x, torch.tensor([900.0, 500.0]), e, p)
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=1e-4, atol=1e-5)
    # blocks must matter (non-degenerate weights): perturbing a late block's gate changes the output
    sd2 = dict(sd)
    sd2["transformer_blocks.1.ff.net.2.weight"] = sd["transformer_blocks.1.ff.net.2.weight"] * 0
    y2 = mmditx_ref.mmdit_forward(sd2, cfg, x, torch.tensor([900.0, 500.0]), e, p)
    assert (y - y2).abs().max() > 1e-3


def test_vae_oracle_self_consistency():
    """[SELF] fixture for the (unpinned) AutoencoderKL decoder restatement + architecture facts that are public:
    the SD3 decoder has 49.5 M parameters in 138 tensors and costs ~10.5 TFLOP per 1024^2 image."""
    from oracle import vae_ref
    z = _load("vae_tiny_self.npz")
    cfg = vae_ref.tiny_config()
    sd = vae_ref.make_synthetic_state_dict(cfg, seed=99)
    lat = torch.randn(1, 16, 4, 4, generator=torch.Generator().manual_seed(98))
    raw = vae_ref.vae_decode(sd, cfg, lat, postprocess=False)
    assert raw.shape == (1, 3, 32, 32)
    np.testing.assert_allclose(raw.numpy(), z["raw"], rtol=1e-4, atol=1e-5)
    img = vae_ref.vae_decode(sd, cfg, lat, quant=lambda t: t.bfloat16().float(), postprocess=True)
    assert np.abs(img.numpy() - z["img_bf16"]).max() <= 2 ** -7   # one bf16 ulp below 1.0
    assert img.min() >= 0 and img.max() <= 1
    shapes = vae_ref.state_dict_shapes(vae_ref.SD3_VAE)
    assert len(shapes) == 138
    assert abs(sum(int(np.prod(s)) for s in shapes.values()) / 49.545e6 - 1) < 1e-3
    assert abs(vae_ref.decode_flops(vae_ref.SD3_VAE, 128, 128) / 1.047e13 - 1) < 2e-3
    # every layer matters: zeroing a late conv changes the image
    sd2 = dict(sd)
    sd2["decoder.up_blocks.3.resnets.1.conv2.weight"] = sd["decoder.up_blocks.3.resnets.1.conv2.weight"] * 0
    assert (vae_ref.vae_decode(sd2, cfg, lat, postprocess=False) - raw).abs().max() > 1e-3


def test_flux_oracle_self_consistency():
    """[SELF] fixture for the (unpinned) FluxTransformer2DModel restatement + public architecture facts: FLUX.1-dev has
    11 901 408 320 parameters; RoPE is a rotation (norm preserving) and the identity on the text tokens (ids = 0)."""
    from oracle import flux_ref as Fx
    z = _load("flux_tiny_self.npz")
    cfg = Fx.tiny_config()
    sd = Fx.make_synthetic_state_dict(cfg, seed=31)
    g = torch.Generator().manual_seed(32)
    B, h, w, Nt = 2, 4, 6, 5
    lat = torch.randn(B, 16, h, w, generator=g)
    x = Fx.pack_latents(lat)
    assert torch.equal(Fx.unpack_latents(x, h, w), lat)
    enc = torch.randn(B, Nt, cfg.joint_attention_dim, generator=g)
    pool = torch.randn(B, cfg.pooled_projection_dim, generator=g)
    ids = Fx.prepare_img_ids(h // 2, w // 2)
    y = Fx.flux_forward(sd, cfg, x, torch.tensor([0.9, 0.4]), torch.tensor([3.5, 3.5]), pool, enc, ids)
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=1e-4, atol=1e-5)
    assert sum(int(np.prod(s)) for s in Fx.state_dict_shapes(Fx.FLUX1_DEV).values()) == 11_901_408_320
    cos, sin = Fx.rope_cos_sin(torch.cat([torch.zeros(Nt, 3), ids], 0))
    v = torch.randn(1, 2, Nt + ids.shape[0], 128, generator=g)
    r = Fx.apply_rope(v, cos, sin)
    torch.testing.assert_close(r.norm(dim=-1), v.norm(dim=-1), rtol=1e-5, atol=1e-5)
    assert torch.equal(r[:, :, :Nt], v[:, :, :Nt])
    # single-stream blocks matter
    sd2 = dict(sd)
    sd2["single_transformer_blocks.1.proj_out.weight"] = sd["single_transformer_blocks.1.proj_out.weight"] * 0
    assert (Fx.flux_forward(sd2, cfg, x, torch.tensor([0.9, 0.4]), torch.tensor([3.5, 3.5]), pool, enc, ids) - y).abs().max() > 1e-4


def test_wan_oracle_self_consistency():
    """[SELF] fixture for the (unpinned) WanTransformer3DModel restatement + checks of its pieces against torch built-ins: the per-frame
    (1,2,2) patchify equals Conv3d, the 3-D rotary embedding is a rotation, un-patchify inverts the token order."""
    from oracle import wan_ref as W
    z = _load("wan_tiny_self.npz")
    cfg = W.tiny_config()
    sd = W.make_synthetic_state_dict(cfg, seed=41)
    g = torch.Generator().manual_seed(42)
    B, T, h, w, Nt = 2, 2, 4, 6, 5
    x = torch.randn(B, 16, T, h, w, generator=g)
    enc = torch.randn(B, Nt, cfg.text_dim, generator=g)
    y = W.wan_forward(sd, cfg, x, torch.tensor([874.0, 249.0]), enc)
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=1e-4, atol=1e-5)
    conv = torch.nn.functional.conv3d(x, sd["patch_embedding.weight"], sd["patch_embedding.bias"], stride=(1, 2, 2)).flatten(2).transpose(1, 2)
    mine = torch.nn.functional.linear(W.patchify(x), sd["patch_embedding.weight"].reshape(cfg.dim, -1), sd["patch_embedding.bias"])
    torch.testing.assert_close(mine, conv, rtol=1e-5, atol=1e-5)
    cos, sin = W.rope_cos_sin(T, h // 2, w // 2)
    v = torch.randn(1, 2, T * (h // 2) * (w // 2), 128, generator=g)
    torch.testing.assert_close(W.apply_rope(v, cos, sin).norm(dim=-1), v.norm(dim=-1), rtol=1e-5, atol=1e-5)
    assert W.rope_axes(128) == (44, 42, 42)
    ts, sig = W.unipc_flow_schedule(4, 3.0)
    assert ts.dtype == torch.int64 and ts.tolist() == sorted(ts.tolist(), reverse=True) and float(sig[-1]) == 0.0


def test_flops_formula():
    # F(4096,333) = 1.125e13, F(256,333) = 9.09e11
    assert abs(mmditx_ref.forward_flops(mmditx_ref.SD35_MEDIUM, 4096, 333) / 1.125e13 - 1) < 2e-3
    assert abs(mmditx_ref.forward_flops(mmditx_ref.SD35_MEDIUM, 256, 333) / 9.09e11 - 1) < 2e-3
