"""TEST INFRASTRUCTURE ONLY.  Runs the REFERENCE's own `SD3_5Adapter.inference()` (imported whole from /root/reference under
`oracle/ref_package.py`) on CPU with `oracle.standin.denoiser` in place of the transformer, and records what it returns: the fixture
`tests/golden/rollout_control_flow.npz` that pins `oracle/rollout_ref.py` (tests/test_rollout_control_flow_pin.py).

    python -m oracle.make_rollout_golden          # build container only (needs /root/reference)
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

from . import ref_package, standin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "rollout_control_flow.npz")
DT = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}

# name: (dynamics, guidance, latent_storage_dtype (None = keep the transformer dtype), N, sde_steps, num_sde_steps, noise_level, eval mode)
CASES = {
    "flow_sde_cfg_fp16": ("Flow-SDE", 4.5, "fp16", 6, [1, 2, 3], 2, 0.7, False),
    "cps_nocfg_bf16": ("CPS", 1.0, "bf16", 5, [0, 1, 2, 3], 3, 0.8, False),
    "dance_cfg_native": ("Dance-SDE", 3.0, None, 5, [1, 2], 1, 0.7, False),
    "eval_ode_cfg_fp16": ("Flow-SDE", 4.5, "fp16", 4, [1, 2, 3], 1, 0.7, True),
}
B, C, H, W, NT, J, P = 2, 16, 64, 64, 7, 128, 128             # image size in pixels (latents 8 x 8)


def _pipeline(tcfg, transformer):
    """Pseudo-pipeline (reference guidance/new_model.md:574-718) with the two diffusers pipeline behaviours `inference()` touches:
    `prepare_latents` (StableDiffusion3Pipeline: one `randn_tensor` of shape (B, C, H/8, W/8) in the requested dtype) and a VAE whose
    decode result is irrelevant here."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _plugin_fakes as F
    from oracle import diffusers_stub as D
    pipe = F.make_pipeline(tcfg, transformer)

    def prepare_latents(batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, int(height) // 8, int(width) // 8)
        return D.randn_tensor(shape, generator=generator, device=device, dtype=dtype)
    pipe.prepare_latents = prepare_latents
    pipe.vae.dtype = torch.float32
    pipe.vae.decode = lambda lat, return_dict=False: (torch.zeros(lat.shape[0], 3, lat.shape[2] * 8, lat.shape[3] * 8),)
    return pipe


def build_sd3(adapter_base=None, dyn="Flow-SDE", storage="fp16", sde_steps=(1, 2, 3), n_sde=1, eta=0.7, is_eval=False, seed=42):
    """The reference's `SD3_5Adapter` (or `adapter_base`: the Flow-Factory plugin class) constructed the way Flow-Factory does, on the
    pseudo-pipeline with the stand-in transformer, in rollout (or eval) mode."""
    ref_package.install()
    sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _plugin_fakes as F
    from flow_factory.hparams import Arguments
    from flow_factory.models.stable_diffusion.sd3_5 import SD3_5Adapter
    from mi355_flow.engine import TransformerConfig
    from mi355_flow.weights import expected_shapes
    cfg = Arguments.load_from_yaml(os.path.join(ref_package.REF_ROOT, "examples/grpo/full/sd3_5/default.yaml"))
    cfg.training_args.latent_storage_dtype = storage
    sa = cfg.scheduler_args
    sa.dynamics_type, sa.noise_level, sa.sde_steps, sa.num_sde_steps, sa.seed = dyn, eta, list(sde_steps), n_sde, seed
    tcfg = TransformerConfig(num_layers=1, num_heads=1, joint_attention_dim=J, pooled_projection_dim=P, pos_embed_max_size=24, dual_layers=())
    tr = F.build_module_tree(expected_shapes(tcfg), cls=F.FakeTransformer).bfloat16()
    tr.forward = lambda hidden_states=None, timestep=None, encoder_hidden_states=None, pooled_projections=None, joint_attention_kwargs=None, \
        return_dict=False: (standin.denoiser(hidden_states, timestep, encoder_hidden_states, pooled_projections),)

    class Ref(adapter_base or SD3_5Adapter):
        def load_pipeline(self):
            return _pipeline(tcfg, tr)

    ad = Ref(cfg, F.FakeAccelerator())
    ad.post_init()
    ad.eval() if is_eval else ad.rollout()
    return ad


def run_reference(case: str, adapter_base=None, explicit_generator=False):
    """-> dict of tensors: the inputs handed to the reference adapter and what its samples hold.  `adapter_base`: run the SAME
    construction and call on another adapter class (the Flow-Factory plugin class with an engine double:
    tests/test_rollout_control_flow_pin.py::test_plugin_host_path_reproduces_the_reference_adapter)."""
    ref_package.install()
    from flow_factory.utils.trajectory_collector import compute_trajectory_indices
    dyn, gs, storage, N, sde_steps, n_sde, eta, is_eval = CASES[case]
    ad = build_sd3(adapter_base, dyn, storage, sde_steps, n_sde, eta, is_eval)
    g = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16()
    pe, pp, ne, npl = mk(B, NT, J), mk(B, P), mk(B, NT, J), mk(B, P)
    seed = 1000 + sorted(CASES).index(case)
    torch.manual_seed(seed)                                  # the reference draws on the global generator (generator=None)
    traj = "all" if is_eval else compute_trajectory_indices(train_timestep_indices=ad.scheduler.train_timesteps, num_inference_steps=N)
    samples = ad.inference(prompt=["p0", "p1"], generator=torch.Generator().manual_seed(77) if explicit_generator else None, height=H, width=W, num_inference_steps=N, guidance_scale=gs, prompt_embeds=pe,
                           pooled_prompt_embeds=pp, negative_prompt_embeds=ne if gs > 1 else None,
                           negative_pooled_prompt_embeds=npl if gs > 1 else None, compute_log_prob=not is_eval, trajectory_indices=traj,
                           extra_call_back_kwargs=["next_latents_mean"])
    sched = ad.scheduler
    out = dict(seed=torch.tensor(seed), is_eval=torch.tensor(int(is_eval)), pe=pe.float(), pp=pp.float(), ne=ne.float(), npl=npl.float(),
               timesteps=samples[0].timesteps.float(), sigmas=sched.sigmas.float(),
               noise_levels=torch.tensor([float(sched.get_noise_level_for_timestep(t)) for t in samples[0].timesteps]),
               all_latents=(torch.stack([s.all_latents for s in samples]).float() if samples[0].all_latents is not None else torch.zeros(0)),
               latents_dtype=torch.tensor({torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}[samples[0].all_latents.dtype] if samples[0].all_latents is not None else -1),
               latent_index_map=(samples[0].latent_index_map if samples[0].latent_index_map is not None else torch.zeros(0)), callback_index_map=samples[0].extra_kwargs["callback_index_map"],
               next_latents_mean=torch.stack([s.extra_kwargs["next_latents_mean"] for s in samples]).float())
    if not is_eval:
        out["log_probs"] = torch.stack([s.log_probs for s in samples]).float()
        out["log_prob_index_map"] = samples[0].log_prob_index_map
        # the optimize() replay of every trained step (trainers/grpo.py:229-263): `t`, `t_next` as (B,) tensors, the stored (x_i, x_{i+1}),
        # `noise_level = scheduler.noise_level`; no-grad here (the values are what matters: ratio == exp(replay - rollout) must be 1)
        lat = torch.stack([s.all_latents for s in samples])
        lmap, ts = samples[0].latent_index_map, samples[0].timesteps
        rep = {k: [] for k in ("log_prob", "noise_pred", "next_latents_mean", "std_dev_t", "dt")}
        with torch.no_grad():
            for i in [int(j) for j in range(N) if samples[0].log_prob_index_map[j] >= 0]:
                t_next = ts[i + 1] if i + 1 < N else torch.tensor(0.0)
                o = ad.forward(t=ts[i].expand(B), t_next=t_next.expand(B), latents=lat[:, lmap[i]], next_latents=lat[:, lmap[i + 1]],
                               prompt_embeds=pe, pooled_prompt_embeds=pp, negative_prompt_embeds=ne if gs > 1 else None,
                               negative_pooled_prompt_embeds=npl if gs > 1 else None, guidance_scale=gs, noise_level=sched.noise_level,
                               compute_log_prob=True, return_kwargs=list(rep))
                for k in rep:
                    rep[k].append(getattr(o, k).float())
        for k, v in rep.items():
            out["replay_" + k] = torch.stack(v)
    return out


# ---------------------------------------------------------------------------------------------- FLUX.1 (models/flux/flux1.py:151-346)
FLUX_CASES = {
    "flux_flow_sde_fp16": ("Flow-SDE", 3.5, "fp16", 5, [1, 2, 3], 2, 0.7),
    "flux_dance_native": ("Dance-SDE", 2.0, None, 4, [0, 1, 2], 1, 0.8),
}


def _flux_pipeline(transformer):
    """Pseudo-pipeline with the FluxPipeline behaviours `Flux1Adapter.inference()` touches, restated from the published pipeline:
    `prepare_latents` = one `randn_tensor` of (B, 16, 2*(H//16), 2*(W//16)) in the requested dtype, packed 2x2 -> (B, h/2*w/2, 64), plus the
    (h/2*w/2, 3) image ids; `_unpack_latents` is its inverse."""
    import torch.nn as nn
    from oracle import diffusers_stub as D
    from oracle import flux_ref as FR
    transformer.config = types.SimpleNamespace(in_channels=64, guidance_embeds=True, num_layers=1, num_single_layers=1, num_attention_heads=1,
                                               attention_head_dim=128, joint_attention_dim=J, pooled_projection_dim=P,
                                               axes_dims_rope=(16, 56, 56))
    vae = nn.Module()
    vae.add_module("decoder", nn.Linear(2, 2))
    vae.config = types.SimpleNamespace(scaling_factor=0.3611, shift_factor=0.1159, latent_channels=16, block_out_channels=(128, 256, 512, 512),
                                       layers_per_block=2, norm_num_groups=32, out_channels=3)
    vae.dtype = torch.float32
    vae.decode = lambda lat, return_dict=False: (torch.zeros(lat.shape[0], 3, lat.shape[2] * 8, lat.shape[3] * 8),)
    pipe = types.SimpleNamespace()
    pipe.transformer, pipe.vae = transformer, vae
    pipe.text_encoder, pipe.text_encoder_2 = nn.Linear(2, 2), nn.Linear(2, 2)
    pipe.tokenizer, pipe.tokenizer_2 = object(), object()
    pipe.vae_scale_factor = 8
    pipe.scheduler = D.FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True, base_shift=0.5,
                                                       max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096)
    pipe.image_processor = types.SimpleNamespace(postprocess=lambda x, output_type="pt": x)
    pipe.maybe_free_model_hooks = lambda: None
    pipe.components = {"transformer": transformer, "vae": vae, "text_encoder": pipe.text_encoder, "text_encoder_2": pipe.text_encoder_2}

    def prepare_latents(batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        h, w = 2 * (int(height) // 16), 2 * (int(width) // 16)
        lat = D.randn_tensor((batch_size, num_channels_latents, h, w), generator=generator, device=device, dtype=dtype)
        return FR.pack_latents(lat), FR.prepare_img_ids(h // 2, w // 2).to(device=device, dtype=dtype)
    pipe.prepare_latents = prepare_latents
    pipe._unpack_latents = lambda lat, height, width, vsf: FR.unpack_latents(lat, 2 * (int(height) // (vsf * 2)), 2 * (int(width) // (vsf * 2)))
    return pipe


def build_flux(case, adapter_base=None, model_level=False):
    """The reference's `Flux1Adapter` (or the plugin class) on the FLUX pseudo-pipeline with the stand-in transformer, in rollout mode."""
    ref_package.install()
    sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _plugin_fakes as F
    from flow_factory.hparams import Arguments
    from flow_factory.models.flux.flux1 import Flux1Adapter
    dyn, gs, storage, N, sde_steps, n_sde, eta = FLUX_CASES[case] if isinstance(case, str) else case
    cfg = Arguments.load_from_yaml(os.path.join(ref_package.REF_ROOT, "examples/grpo/full/flux1/default.yaml"))
    cfg.training_args.latent_storage_dtype = storage
    sa = cfg.scheduler_args
    sa.dynamics_type, sa.noise_level, sa.sde_steps, sa.num_sde_steps, sa.seed = dyn, eta, list(sde_steps), n_sde, 42
    shapes = {"transformer_blocks.0.attn.to_q.weight": (8, 8), "transformer_blocks.0.attn.to_q.bias": (8,), "x_embedder.weight": (8, 8)}
    tr = F.build_module_tree(shapes, buffers=(), cls=F.FakeTransformer).bfloat16()
    tr.forward = lambda hidden_states=None, timestep=None, guidance=None, pooled_projections=None, encoder_hidden_states=None, txt_ids=None, \
        img_ids=None, joint_attention_kwargs=None, return_dict=False: (
            (standin.flux_transformer_call if model_level else standin.flux_denoiser)(
                hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids),)

    class Ref(adapter_base or Flux1Adapter):
        def load_pipeline(self):
            return _flux_pipeline(tr)

    ad = Ref(cfg, F.FakeAccelerator())
    ad.post_init()
    ad.rollout()
    return ad


def run_reference_flux(case, adapter_base=None, callbacks=True, explicit_generator=False, traj="train", clp=True, seed=None, model_level=False):
    ref_package.install()
    from flow_factory.utils.trajectory_collector import compute_trajectory_indices
    dyn, gs, storage, N, sde_steps, n_sde, eta = FLUX_CASES[case] if isinstance(case, str) else case
    ad = build_flux(case, adapter_base, model_level)
    g = torch.Generator().manual_seed(21)
    pe, pp = torch.randn(B, NT, J, generator=g).bfloat16(), torch.randn(B, P, generator=g).bfloat16()
    seed = (2000 + sorted(FLUX_CASES).index(case)) if seed is None else seed
    torch.manual_seed(seed)
    if traj == "train":
        traj = compute_trajectory_indices(train_timestep_indices=ad.scheduler.train_timesteps, num_inference_steps=N)
    samples = ad.inference(prompt=["p0", "p1"], generator=torch.Generator().manual_seed(77) if explicit_generator else None, height=H, width=W, num_inference_steps=N, guidance_scale=gs, prompt_embeds=pe,
                           pooled_prompt_embeds=pp, compute_log_prob=clp, trajectory_indices=traj,
                           extra_call_back_kwargs=["next_latents_mean"] if callbacks else [])
    sched = ad.scheduler
    if not callbacks:
        for s_ in samples:
            s_.extra_kwargs["next_latents_mean"] = torch.zeros(0)
    return dict(seed=torch.tensor(seed), guidance=torch.tensor(gs), pe=pe.float(), pp=pp.float(), timesteps=samples[0].timesteps.float(),
                sigmas=sched.sigmas.float(), noise_levels=torch.tensor([float(sched.get_noise_level_for_timestep(t)) for t in samples[0].timesteps]),
                all_latents=(torch.stack([s.all_latents for s in samples]).float() if samples[0].all_latents is not None else torch.zeros(0)), img_ids=samples[0].img_ids.float(),
                latents_dtype=torch.tensor({torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}[samples[0].all_latents.dtype] if samples[0].all_latents is not None else -1),
                latent_index_map=(samples[0].latent_index_map if samples[0].latent_index_map is not None else torch.zeros(0)), callback_index_map=samples[0].extra_kwargs["callback_index_map"],
                next_latents_mean=torch.stack([s.extra_kwargs["next_latents_mean"] for s in samples]).float(),
                **({} if samples[0].log_probs is None else dict(log_probs=torch.stack([s.log_probs for s in samples]).float(),
                                                                 log_prob_index_map=samples[0].log_prob_index_map)))


# ---------------------------------------------------------------------------------------------- Qwen-Image (models/qwen_image/qwen_image.py:288-600)
QWEN_CASES = {
    "qwen_flow_sde_cfg_ragged": ("Flow-SDE", 4.0, None, 5, [1, 2, 3], 2, 0.7),
    "qwen_cps_nocfg_fp16": ("CPS", 1.0, "fp16", 4, [0, 1, 2], 1, 0.8),
}
QJ = 64


def build_qwen(case, adapter_base=None, model_level=False):
    """The reference's `QwenImageAdapter` (or the plugin class) on the Qwen pseudo-pipeline with the stand-in transformer, in rollout mode."""
    ref_package.install()
    sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _plugin_fakes as F
    import mi355_flow.qwen as QW
    from contextlib import nullcontext
    from flow_factory.hparams import Arguments
    from flow_factory.models.qwen_image.qwen_image import QwenImageAdapter
    from oracle import diffusers_stub as D
    from oracle import flux_ref as FR
    dyn, gs, storage, N, sde_steps, n_sde, eta = QWEN_CASES[case] if isinstance(case, str) else case
    cfg = Arguments.load_from_yaml(os.path.join(ref_package.REF_ROOT, "examples/grpo/full/qwen_image/default.yaml"))
    cfg.training_args.latent_storage_dtype = storage
    sa = cfg.scheduler_args
    sa.dynamics_type, sa.noise_level, sa.sde_steps, sa.num_sde_steps, sa.seed = dyn, eta, list(sde_steps), n_sde, 42
    tcfg = QW.QwenConfig(num_layers=1, num_attention_heads=1, joint_attention_dim=QJ)
    tr = F.build_module_tree({"transformer_blocks.0.attn.to_q.weight": (8, 8), "transformer_blocks.0.attn.to_q.bias": (8,)}, buffers=(),
                             cls=F.FakeTransformer).bfloat16()
    tr.forward = lambda hidden_states=None, timestep=None, guidance=None, encoder_hidden_states_mask=None, encoder_hidden_states=None, \
        img_shapes=None, txt_seq_lens=None, attention_kwargs=None, return_dict=False: (
            (standin.qwen_transformer_call if model_level else standin.qwen_denoiser)(
                hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_mask, img_shapes, txt_seq_lens),)
    tr.cache_context = lambda name: nullcontext()

    def prepare_latents(batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        # QwenImagePipeline.prepare_latents: one randn of (B, 1, 16, h, w), packed 2x2 like FLUX on the single frame
        h, w = 2 * (int(height) // 16), 2 * (int(width) // 16)
        lat = D.randn_tensor((batch_size, 1, num_channels_latents, h, w), generator=generator, device=device, dtype=dtype)
        return FR.pack_latents(lat[:, 0])

    class Ref(adapter_base or QwenImageAdapter):
        def load_pipeline(self):
            pipe = F.make_qwen_pipeline(tcfg, tr)
            pipe.prepare_latents = prepare_latents
            pipe._unpack_latents = lambda lat, height, width, vsf: FR.unpack_latents(lat, 2 * (int(height) // 16), 2 * (int(width) // 16)).unsqueeze(2)
            pipe.vae.dtype = torch.float32
            pipe.vae.decode = lambda lat, return_dict=False: (torch.zeros(lat.shape[0], 3, 1, lat.shape[-2] * 8, lat.shape[-1] * 8),)
            return pipe

    ad = Ref(cfg, F.FakeAccelerator())
    ad.post_init()
    ad.rollout()
    return ad


def run_reference_qwen(case, adapter_base=None, callbacks=True, explicit_generator=False, traj="train", clp=True, seed=None, model_level=False):
    ref_package.install()
    from flow_factory.utils.trajectory_collector import compute_trajectory_indices
    dyn, gs, storage, N, sde_steps, n_sde, eta = QWEN_CASES[case] if isinstance(case, str) else case
    ad = build_qwen(case, adapter_base, model_level)
    g = torch.Generator().manual_seed(31)
    lens, nlens = [5, 9], [3, 3]
    pe = [torch.randn(n, QJ, generator=g).bfloat16() for n in lens]
    ne = [torch.randn(n, QJ, generator=g).bfloat16() for n in nlens]
    seed = (3000 + sorted(QWEN_CASES).index(case)) if seed is None else seed
    torch.manual_seed(seed)
    if traj == "train":
        traj = compute_trajectory_indices(train_timestep_indices=ad.scheduler.train_timesteps, num_inference_steps=N)
    samples = ad.inference(prompt=["p0", "p1"], generator=torch.Generator().manual_seed(77) if explicit_generator else None, height=H, width=W, num_inference_steps=N, guidance_scale=gs, prompt_embeds=pe,
                           prompt_embeds_mask=[torch.ones(n, dtype=torch.long) for n in lens], prompt_ids=[torch.arange(n) for n in lens],
                           negative_prompt_embeds=ne if gs > 1 else None,
                           negative_prompt_embeds_mask=[torch.ones(n, dtype=torch.long) for n in nlens] if gs > 1 else None,
                           compute_log_prob=clp, trajectory_indices=traj, extra_call_back_kwargs=["next_latents_mean"] if callbacks else [])
    sched = ad.scheduler
    if not callbacks:
        for s_ in samples:
            s_.extra_kwargs["next_latents_mean"] = torch.zeros(0)
    pad = lambda seq: torch.nn.utils.rnn.pad_sequence([x.float() for x in seq], batch_first=True)      # noqa: E731
    return dict(seed=torch.tensor(seed), guidance=torch.tensor(gs), pe=pad(pe), ne=pad(ne), lens=torch.tensor(lens), nlens=torch.tensor(nlens),
                timesteps=samples[0].timesteps.float(), sigmas=sched.sigmas.float(),
                noise_levels=torch.tensor([float(sched.get_noise_level_for_timestep(t)) for t in samples[0].timesteps]),
                all_latents=(torch.stack([s.all_latents for s in samples]).float() if samples[0].all_latents is not None else torch.zeros(0)), hw=torch.tensor([H // 16, W // 16]),
                latents_dtype=torch.tensor({torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}[samples[0].all_latents.dtype] if samples[0].all_latents is not None else -1),
                latent_index_map=(samples[0].latent_index_map if samples[0].latent_index_map is not None else torch.zeros(0)), callback_index_map=samples[0].extra_kwargs["callback_index_map"],
                next_latents_mean=torch.stack([s.extra_kwargs["next_latents_mean"] for s in samples]).float(),
                **({} if samples[0].log_probs is None else dict(log_probs=torch.stack([s.log_probs for s in samples]).float(),
                                                                 log_prob_index_map=samples[0].log_prob_index_map)))


# ---------------------------------------------------------------------------------------------- Wan2.1 / Wan2.2 T2V (models/wan/wan2_t2v.py:234-543)
# name: (dynamics, guidance, guidance_2, boundary_ratio (None = single transformer), storage, N, sde_steps, num_sde_steps, noise_level)
WAN_CASES = {
    "wan21_flow_sde_cfg_fp16": ("Flow-SDE", 5.0, None, None, "fp16", 5, [1, 2, 3], 2, 0.7),
    "wan22_two_expert_cps": ("CPS", 4.0, 3.0, 0.6, "bf16", 6, [0, 1, 2, 3, 4], 3, 0.8),
}
WAN_FRAMES, WAN_TD = 5, 96          # 5 frames -> 2 latent frames; text width
# live-only cases (not in the committed fixture): Wan2.2-TI2V-5B in text-to-video use -- 48 latent channels, 4 x 16 x 16 VAE compression,
# `expand_timesteps` (one timestep per token, all equal): tests/test_rollout_control_flow_pin.py compares the plugin with the reference adapter
WAN_LIVE_CASES = {"wan22_ti2v_expand_timesteps": ("Flow-SDE", 5.0, None, None, "bf16", 5, [1, 2, 3], 2, 0.7)}
WAN_GEOMETRY = {"wan22_ti2v_expand_timesteps": dict(z=48, spatial=16, expand=True)}


def _wan_case(case):
    return ({**WAN_CASES, **WAN_LIVE_CASES}[case] if isinstance(case, str) else case)


def build_wan(case, adapter_base=None):
    """The reference's `Wan2_T2V_Adapter` (or the plugin class) on the Wan pseudo-pipeline with the stand-in transformer(s), in rollout mode;
    returns (adapter, N)."""
    ref_package.install()
    sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _plugin_fakes as F
    import torch.nn as nn
    from contextlib import nullcontext
    from flow_factory.hparams import Arguments
    from flow_factory.models.wan.wan2_t2v import Wan2_T2V_Adapter
    from flow_factory.utils.trajectory_collector import compute_trajectory_indices
    from oracle import diffusers_stub as D
    dyn, gs, gs2, ratio, storage, N, sde_steps, n_sde, eta = _wan_case(case)
    geo = WAN_GEOMETRY.get(case if isinstance(case, str) else "", dict(z=16, spatial=8, expand=False))
    cfg = Arguments.load_from_yaml(os.path.join(ref_package.REF_ROOT, "examples/grpo/full/wan21/t2v.yaml"))
    cfg.training_args.latent_storage_dtype = storage
    sa = cfg.scheduler_args
    sa.dynamics_type, sa.noise_level, sa.sde_steps, sa.num_sde_steps, sa.seed = dyn, eta, list(sde_steps), n_sde, 42

    def transformer(expert):
        tr = F.build_module_tree({"blocks.0.attn1.to_q.weight": (8, 8), "blocks.0.attn1.to_q.bias": (8,)}, buffers=(), cls=F.FakeTransformer).bfloat16()
        tr.config = types.SimpleNamespace(in_channels=geo["z"], out_channels=geo["z"], patch_size=(1, 2, 2), num_layers=1, num_attention_heads=1,
                                          attention_head_dim=128, ffn_dim=64, text_dim=WAN_TD, freq_dim=256, eps=1e-6)
        tr.forward = lambda hidden_states=None, timestep=None, encoder_hidden_states=None, attention_kwargs=None, return_dict=False: (
            standin.wan_denoiser(hidden_states, timestep, encoder_hidden_states, expert),)
        tr.cache_context = lambda name: nullcontext()
        return tr

    def pipeline():
        vae = nn.Module()
        vae.add_module("decoder", nn.Linear(2, 2))
        Z, SP = geo["z"], geo["spatial"]
        vae.config = types.SimpleNamespace(z_dim=Z, base_dim=96, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True),
                                           latents_mean=[0.0] * Z, latents_std=[1.0] * Z)
        vae.dtype = torch.float32
        vae.decode = lambda lat, return_dict=False: (torch.zeros(lat.shape[0], 3, 1 + 4 * (lat.shape[2] - 1), lat.shape[3] * SP, lat.shape[4] * SP),)
        pipe = types.SimpleNamespace()
        pipe.transformer, pipe.vae = transformer(0), vae
        pipe.transformer_2 = transformer(1) if ratio is not None else None
        pipe.text_encoder, pipe.tokenizer = nn.Linear(2, 2), object()
        pipe.vae_scale_factor_temporal, pipe.vae_scale_factor_spatial = 4, SP
        pipe.config = types.SimpleNamespace(boundary_ratio=ratio, expand_timesteps=geo["expand"])
        pipe.scheduler = D.UniPCMultistepScheduler(num_train_timesteps=1000, use_flow_sigmas=True, flow_shift=3.0)
        pipe.video_processor = types.SimpleNamespace(postprocess_video=lambda v, output_type="pt": v)
        pipe.maybe_free_model_hooks = lambda: None
        pipe.components = {"transformer": pipe.transformer, "vae": vae, "text_encoder": pipe.text_encoder}
        if pipe.transformer_2 is not None:
            pipe.components["transformer_2"] = pipe.transformer_2

        def prepare_latents(batch_size, num_channels_latents, height, width, num_frames, dtype, device, generator, latents=None):
            # WanPipeline.prepare_latents: one randn of (B, 16, (F - 1) // 4 + 1, H / 8, W / 8) in the requested dtype (fp32 here)
            shape = (batch_size, num_channels_latents, (int(num_frames) - 1) // 4 + 1, int(height) // SP, int(width) // SP)
            return D.randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        pipe.prepare_latents = prepare_latents
        return pipe

    class Ref(adapter_base or Wan2_T2V_Adapter):
        def load_pipeline(self):
            return pipeline()

    ad = Ref(cfg, F.FakeAccelerator())
    ad.post_init()
    ad.rollout()
    return ad, N


def run_reference_wan(case, adapter_base=None, callbacks=True, explicit_generator=False, traj="train", clp=True, seed=None, evaluation=False):
    ref_package.install()
    from flow_factory.utils.trajectory_collector import compute_trajectory_indices
    dyn, gs, gs2, ratio, storage, N, sde_steps, n_sde, eta = _wan_case(case)
    ad, N = build_wan(case, adapter_base)
    g = torch.Generator().manual_seed(41)
    pe, ne = torch.randn(B, NT, WAN_TD, generator=g).bfloat16(), torch.randn(B, NT, WAN_TD, generator=g).bfloat16()
    if seed is None:          # (the live-only cases draw from their own range: the committed fixtures keep the seeds they were generated with)
        seed = 4000 + sorted(WAN_CASES).index(case) if case in WAN_CASES else 4100 + sorted(WAN_LIVE_CASES).index(case)
    torch.manual_seed(seed)
    ad.scheduler.set_timesteps(N)                      # train_timesteps (the SDE-step selection) needs a schedule
    if traj == "train":
        traj = compute_trajectory_indices(train_timestep_indices=ad.scheduler.train_timesteps, num_inference_steps=N)
    if evaluation:                                     # models/abc.py:351-378: eval() puts the scheduler into its `is_eval` branch
        ad.eval()
    samples = ad.inference(prompt=["p0", "p1"], generator=torch.Generator().manual_seed(77) if explicit_generator else None, negative_prompt=["", ""], height=H, width=W, num_frames=WAN_FRAMES, num_inference_steps=N,
                           guidance_scale=gs, guidance_scale_2=gs2, prompt_ids=torch.zeros(B, 4, dtype=torch.long), prompt_embeds=pe,
                           negative_prompt_ids=torch.zeros(B, 4, dtype=torch.long), negative_prompt_embeds=ne, compute_log_prob=clp,
                           trajectory_indices=traj, extra_call_back_kwargs=["next_latents_mean"] if callbacks else [])
    sched = ad.scheduler
    if not callbacks:
        for s_ in samples:
            s_.extra_kwargs["next_latents_mean"] = torch.zeros(0)
    return dict(seed=torch.tensor(seed), guidance=torch.tensor(gs), guidance_2=torch.tensor(gs2 if gs2 is not None else -1.0),
                boundary_timestep=torch.tensor(ratio * 1000.0 if ratio is not None else -1.0), pe=pe.float(), ne=ne.float(),
                timesteps=samples[0].timesteps.long(), sigmas=sched.sigmas.float(),
                noise_levels=torch.tensor([float(sched.get_noise_level_for_timestep(t)) for t in samples[0].timesteps]),
                all_latents=(torch.stack([s.all_latents for s in samples]).float() if samples[0].all_latents is not None else torch.zeros(0)),
                latents_dtype=torch.tensor({torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}[samples[0].all_latents.dtype] if samples[0].all_latents is not None else -1),
                latent_index_map=(samples[0].latent_index_map if samples[0].latent_index_map is not None else torch.zeros(0)), callback_index_map=samples[0].extra_kwargs["callback_index_map"],
                next_latents_mean=torch.stack([s.extra_kwargs["next_latents_mean"] for s in samples]).float(),
                **({} if samples[0].log_probs is None else dict(log_probs=torch.stack([s.log_probs for s in samples]).float(),
                                                                 log_prob_index_map=samples[0].log_prob_index_map)))


def main():
    blob = {}
    for case in CASES:
        for k, v in run_reference(case).items():
            blob[f"{case}/{k}"] = v.detach().cpu().numpy()
    for case in FLUX_CASES:
        for k, v in run_reference_flux(case).items():
            blob[f"{case}/{k}"] = v.detach().cpu().numpy()
    for case in QWEN_CASES:
        for k, v in run_reference_qwen(case).items():
            blob[f"{case}/{k}"] = v.detach().cpu().numpy()
    for case in WAN_CASES:
        for k, v in run_reference_wan(case).items():
            blob[f"{case}/{k}"] = v.detach().cpu().numpy()
    np.savez_compressed(OUT, **blob)
    print(f"wrote {OUT}: {len(blob)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
