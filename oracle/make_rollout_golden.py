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


def run_reference(case: str):
    """-> dict of tensors: the inputs handed to the reference adapter and what its samples hold."""
    ref_package.install()
    sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _plugin_fakes as F
    from flow_factory.hparams import Arguments
    from flow_factory.models.stable_diffusion.sd3_5 import SD3_5Adapter
    from flow_factory.utils.trajectory_collector import compute_trajectory_indices
    from mi355_flow.engine import TransformerConfig
    from mi355_flow.weights import expected_shapes
    dyn, gs, storage, N, sde_steps, n_sde, eta, is_eval = CASES[case]
    cfg = Arguments.load_from_yaml(os.path.join(ref_package.REF_ROOT, "examples/grpo/full/sd3_5/default.yaml"))
    cfg.training_args.latent_storage_dtype = storage
    sa = cfg.scheduler_args
    sa.dynamics_type, sa.noise_level, sa.sde_steps, sa.num_sde_steps, sa.seed = dyn, eta, list(sde_steps), n_sde, 42
    tcfg = TransformerConfig(num_layers=1, num_heads=1, joint_attention_dim=J, pooled_projection_dim=P, pos_embed_max_size=24, dual_layers=())
    tr = F.build_module_tree(expected_shapes(tcfg), cls=F.FakeTransformer).bfloat16()
    tr.forward = lambda hidden_states=None, timestep=None, encoder_hidden_states=None, pooled_projections=None, joint_attention_kwargs=None, \
        return_dict=False: (standin.denoiser(hidden_states, timestep, encoder_hidden_states, pooled_projections),)

    class Ref(SD3_5Adapter):
        def load_pipeline(self):
            return _pipeline(tcfg, tr)

    ad = Ref(cfg, F.FakeAccelerator())
    ad.post_init()
    ad.eval() if is_eval else ad.rollout()
    g = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16()
    pe, pp, ne, npl = mk(B, NT, J), mk(B, P), mk(B, NT, J), mk(B, P)
    seed = 1000 + sorted(CASES).index(case)
    torch.manual_seed(seed)                                  # the reference draws on the global generator (generator=None)
    traj = "all" if is_eval else compute_trajectory_indices(train_timestep_indices=ad.scheduler.train_timesteps, num_inference_steps=N)
    samples = ad.inference(prompt=["p0", "p1"], height=H, width=W, num_inference_steps=N, guidance_scale=gs, prompt_embeds=pe,
                           pooled_prompt_embeds=pp, negative_prompt_embeds=ne if gs > 1 else None,
                           negative_pooled_prompt_embeds=npl if gs > 1 else None, compute_log_prob=not is_eval, trajectory_indices=traj,
                           extra_call_back_kwargs=["next_latents_mean"])
    sched = ad.scheduler
    out = dict(seed=torch.tensor(seed), is_eval=torch.tensor(int(is_eval)), pe=pe.float(), pp=pp.float(), ne=ne.float(), npl=npl.float(),
               timesteps=samples[0].timesteps.float(), sigmas=sched.sigmas.float(),
               noise_levels=torch.tensor([float(sched.get_noise_level_for_timestep(t)) for t in samples[0].timesteps]),
               all_latents=torch.stack([s.all_latents for s in samples]).float(),
               latents_dtype=torch.tensor({torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}[samples[0].all_latents.dtype]),
               latent_index_map=samples[0].latent_index_map, callback_index_map=samples[0].extra_kwargs["callback_index_map"],
               next_latents_mean=torch.stack([s.extra_kwargs["next_latents_mean"] for s in samples]).float())
    if not is_eval:
        out["log_probs"] = torch.stack([s.log_probs for s in samples]).float()
        out["log_prob_index_map"] = samples[0].log_prob_index_map
    return out


def main():
    blob = {}
    for case in CASES:
        for k, v in run_reference(case).items():
            blob[f"{case}/{k}"] = v.detach().cpu().numpy()
    np.savez_compressed(OUT, **blob)
    print(f"wrote {OUT}: {len(blob)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
