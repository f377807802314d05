"""`SD3_5NativeAdapter`: the rollout hot path of Flow-Factory's `SD3_5Adapter` on the MI355X engine.

Mirrors `SD3_5Adapter.inference` / `SD3_5Adapter.forward` (reference
src/flow_factory/models/stable_diffusion/sd3_5.py:176-349, :352-448): the parameter names and
defaults are the ABI (the trainer filters kwargs by signature, trainers/grpo.py:165,252), the RNG
draw order, latent storage dtype, CFG batch order and collector semantics are the reference's.
The N-step loop itself runs inside libmi355flow.so (`mi355_rollout`) with no host sync.

This class is self-contained (no `flow_factory` / `diffusers` import) so it runs and is tested
here; `mi355_flow.flow_factory_plugin` subclasses the reference's own `SD3_5Adapter` with it when
Flow-Factory is installed.
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from .engine import Engine, TransformerConfig
from .samples import SD3_5Sample
from .scheduler import (FlowMatchEulerDiscreteSDEScheduler, SDESchedulerOutput, host_noise_levels, randn_tensor,
                        set_scheduler_timesteps)
from .trajectory import TrajectoryIndicesType, _resolve, collect_rollout

logger = logging.getLogger(__name__)

_DTYPE_MAP = {"fp16": torch.float16, "float16": torch.float16, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
              "fp32": torch.float32, "float32": torch.float32}
VAE_SCALE_FACTOR = 8  # SD3 VAE: latent = image / 8


class NativeRolloutMixin:
    """inference()/forward() on the engine.  Host classes provide: `engine` (Engine), `scheduler`
    (FlowMatchEulerDiscreteSDEScheduler-like), `device`, `transformer_dtype`,
    `latent_storage_dtype`, `decode_latents(latents, output_type)`."""

    engine: Engine
    scheduler: FlowMatchEulerDiscreteSDEScheduler
    # hooks the Flow-Factory plugin overrides with the reference's own objects (flow_factory_plugin.py): the sample class the
    # trainer stacks (`BaseSample.stack`, trainers/grpo.py:215), the schedule setter and the scheduler-output container
    _sample_cls = SD3_5Sample
    _output_cls = SDESchedulerOutput
    _set_timesteps = staticmethod(set_scheduler_timesteps)

    def _before_engine_call(self) -> None:
        """Hook run at the top of inference() / forward(): the Flow-Factory plugin re-binds changed weights here."""

    def _grad_fallback(self, why: str, kwargs: Dict[str, Any]):
        """Grad-mode forward() with trainable parameters the native backward does not cover.  The Flow-Factory plugin overrides this
        with the reference's autograd path; standalone there is no other implementation to fall back to."""
        raise NotImplementedError(f"mi355_flow: grad-mode forward() is not available: {why}")

    def _check_joint_attention_kwargs(self, jak: Optional[Dict[str, Any]]) -> None:
        """`joint_attention_kwargs` (sd3_5.py:421-428 forwards them to the transformer): diffusers consumes `scale` (LoRA scale of
        the peft backend: `scale_lora_layers`) and IP-adapter inputs.  The engine honours `scale` through the weight binding (LoRA
        deltas are merged with that extra factor; no effect without LoRA layers, as in diffusers); anything else raises."""
        jak = dict(jak or {})
        scale = float(jak.pop("scale", 1.0))
        if jak:
            raise NotImplementedError(f"mi355_flow: joint_attention_kwargs {sorted(jak)} are not supported by the native engine "
                                      "(only the LoRA `scale` is)")
        live = getattr(self, "_live_weights", None)
        if live is not None:
            live.set_lora_scale(scale)
        elif scale != 1.0:
            logger.warning("joint_attention_kwargs['scale'] has no effect: no LoRA layers are bound to the native engine")

    # -------------------------------------------------------------- latent casting (models/abc.py:172-182)
    def cast_latents(self, latents: torch.Tensor, default_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        target = self.latent_storage_dtype or default_dtype
        if target is None or latents.dtype == target:
            return latents
        if target == torch.float16:
            latents = latents.clamp(-65504.0, 65504.0)  # unconditional: no abs().max().item() host sync
        return latents.to(target)

    # -------------------------------------------------------------- rollout
    @torch.no_grad()
    def inference(
        self,
        prompt: Union[str, List[str], None] = None,
        negative_prompt: Optional[Union[str, List[str]]] = None,
        height: Optional[int] = 1024,
        width: Optional[int] = 1024,
        num_inference_steps: Optional[int] = 50,
        guidance_scale: float = 7.5,
        generator: Optional[torch.Generator] = None,
        joint_attention_kwargs: Optional[Dict[str, Any]] = None,
        prompt_ids: Optional[torch.Tensor] = None,
        prompt_embeds: Optional[torch.Tensor] = None,
        pooled_prompt_embeds: Optional[torch.Tensor] = None,
        negative_prompt_ids: Optional[torch.Tensor] = None,
        negative_prompt_embeds: Optional[torch.Tensor] = None,
        negative_pooled_prompt_embeds: Optional[torch.Tensor] = None,
        compute_log_prob: bool = True,
        extra_call_back_kwargs: List[str] = [],
        trajectory_indices: TrajectoryIndicesType = "all",
    ) -> List[SD3_5Sample]:
        self._before_engine_call()
        device = self.device
        dtype = self.transformer_dtype
        self._check_joint_attention_kwargs(joint_attention_kwargs)
        do_cfg = guidance_scale > 1.0
        has_neg = negative_prompt is not None or (negative_prompt_embeds is not None and negative_pooled_prompt_embeds is not None)
        if do_cfg and not has_neg:
            logger.warning("No negative prompt/embeds provided, classifier-free-guidance will be disabled.")
            do_cfg = False
        if prompt_embeds is None or pooled_prompt_embeds is None:
            enc = self.encode_prompt(prompt, negative_prompt, guidance_scale=guidance_scale, device=device)
            prompt_embeds, pooled_prompt_embeds, prompt_ids = enc["prompt_embeds"], enc["pooled_prompt_embeds"], enc["prompt_ids"]
            if do_cfg:
                negative_prompt_embeds = enc["negative_prompt_embeds"]
                negative_prompt_ids = enc["negative_prompt_ids"]
                negative_pooled_prompt_embeds = enc["negative_pooled_prompt_embeds"]
        else:
            prompt_embeds, pooled_prompt_embeds = prompt_embeds.to(device), pooled_prompt_embeds.to(device)
            if do_cfg:
                if negative_prompt_embeds is None or negative_pooled_prompt_embeds is None:
                    raise ValueError("classifier-free guidance with pre-encoded prompts needs negative_prompt_embeds "
                                     "and negative_pooled_prompt_embeds")
                negative_prompt_embeds = negative_prompt_embeds.to(device)
                negative_pooled_prompt_embeds = negative_pooled_prompt_embeds.to(device)
        B = len(prompt_embeds)
        C = self.engine.cfg.in_channels
        h, w = int(height) // VAE_SCALE_FACTOR, int(width) // VAE_SCALE_FACTOR
        N = int(num_inference_steps)

        # RNG, in the reference's order: prepare_latents on `generator` (transformer dtype; None = the global device generator, a CPU
        # generator draws on the host like diffusers' randn_tensor), then one fp32 draw per step -- also when noise_level == 0, but none
        # at all under ODE dynamics (flow_match_euler_discrete.py:329-340 draws nothing).  The step draws ALWAYS come from the global
        # generator: the reference hands `generator` to `prepare_latents` only (sd3_5.py:242-250), never to `scheduler.step` (:436-446)
        # (found by tests/test_rollout_control_flow_pin.py::test_differential_sweep_plugin_vs_reference_adapter)
        dyn = self.scheduler.dynamics_type
        latents = randn_tensor((B, C, h, w), generator=generator, device=device, dtype=dtype)
        step_noise = None
        if dyn != "ODE":
            step_noise = torch.empty((N, B, C, h, w), device=device, dtype=torch.float32)
            for i in range(N):
                step_noise[i] = randn_tensor((B, C, h, w), generator=None, device=device, dtype=torch.float32)

        ps = self.engine.cfg.patch_size
        timesteps = self._set_timesteps(self.scheduler, N, seq_len=(h // ps) * (w // ps), device=device)
        ts_host = [float(t) for t in timesteps.tolist()]          # one D2H before the loop, none inside
        sig_host = [float(s) for s in self.scheduler.sigmas.tolist()]
        eta_host = host_noise_levels(self.scheduler, N)
        storage = self.latent_storage_dtype or dtype

        n_text = prompt_embeds.shape[1]
        plan = self.engine.plan(B, 2 if do_cfg else 1, h, w, n_text, N)
        stepwise = any(k != "noise_level" for k in extra_call_back_kwargs)
        kept = _resolve(trajectory_indices, N + 1)
        keep_positions = list(range(N + 1)) if kept is None else sorted(kept)
        if not stepwise:
            lat_kept, log_probs, final = plan.rollout(
                ts_host, sig_host, eta_host, dyn, guidance_scale, latents, storage, step_noise,
                prompt_embeds, pooled_prompt_embeds, negative_prompt_embeds if do_cfg else None,
                negative_pooled_prompt_embeds if do_cfg else None, keep_positions=keep_positions,
                compute_log_prob=compute_log_prob)
            pos_to_slot = {p: s for s, p in enumerate(keep_positions)}
            get_lat = lambda pos: lat_kept[pos_to_slot[pos]]
            step_outputs = None
        else:
            all_lat, log_probs, step_outputs = self._rollout_stepwise(
                plan, ts_host, sig_host, eta_host, guidance_scale, latents, storage, step_noise, prompt_embeds,
                pooled_prompt_embeds, negative_prompt_embeds if do_cfg else None,
                negative_pooled_prompt_embeds if do_cfg else None, compute_log_prob, extra_call_back_kwargs)
            get_lat = lambda pos: all_lat[pos]
            final = all_lat[N]

        # collectors: exactly the reference's bookkeeping (sd3_5.py:265-304) over the engine outputs
        traj = collect_rollout(trajectory_indices, N, get_lat, log_probs, eta_host, compute_log_prob, step_outputs, extra_call_back_kwargs,
                               captured_noise_levels=host_noise_levels(self.scheduler, N, effective=False), dynamics=dyn)
        images = self.decode_latents(latents=final, output_type="pt")
        samples = []
        for b in range(B):
            samples.append(self._sample_cls(
                timesteps=timesteps,
                **traj.per_sample(b),
                prompt=prompt[b] if isinstance(prompt, list) else prompt,
                prompt_ids=prompt_ids[b] if prompt_ids is not None else None,
                prompt_embeds=prompt_embeds[b],
                pooled_prompt_embeds=pooled_prompt_embeds[b],
                negative_prompt=(negative_prompt[b] if isinstance(negative_prompt, list) else negative_prompt),
                negative_prompt_ids=negative_prompt_ids[b] if negative_prompt_ids is not None else None,
                negative_prompt_embeds=negative_prompt_embeds[b] if negative_prompt_embeds is not None else None,
                negative_pooled_prompt_embeds=negative_pooled_prompt_embeds[b] if negative_pooled_prompt_embeds is not None else None,
                height=height, width=width,
                image=images[b] if images is not None else None,
            ))
        return samples

    def _rollout_stepwise(self, plan, ts, sig, eta, guidance, latents, storage, step_noise, pe, pp, ne, npl, compute_log_prob,
                          extra_keys):
        """Per-step engine calls (still all-HIP) for rollouts that ask for per-step callback tensors."""
        N, B = len(ts), latents.shape[0]
        cur = self.cast_latents(latents, storage)
        all_lat = [cur]
        log_probs = torch.full((N, B), float("nan"), device=latents.device)
        outs = []
        want = tuple(k for k in extra_keys if k in ("next_latents", "next_latents_mean", "noise_pred", "std_dev_t", "dt"))
        for i in range(N):
            t_next = ts[i + 1] if i + 1 < N else 0.0
            clp = compute_log_prob and eta[i] > 0
            enc_a, pool_a, enc_b, pool_b = (ne, npl, pe, pp) if ne is not None else (pe, pp, None, None)
            o = plan.denoise_step(cur, torch.tensor(ts[i]), enc_a, pool_a, enc_b, pool_b, guidance,
                                  torch.tensor(ts[i]) / 1000, torch.tensor(t_next) / 1000, eta[i], sig[1],
                                  self.scheduler.dynamics_type, noise=step_noise[i] if step_noise is not None else None,
                                  compute_log_prob=clp, want=want)
            if clp:
                log_probs[i] = o.log_prob
            cur = o.next_storage
            all_lat.append(cur)
            outs.append(o)
        return all_lat, log_probs, outs

    # -------------------------------------------------------------- one step (rollout or replay)
    def forward(
        self,
        t: torch.Tensor,
        latents: torch.Tensor,
        prompt_embeds: torch.Tensor,
        pooled_prompt_embeds: torch.Tensor,
        negative_prompt_embeds: Optional[torch.Tensor] = None,
        negative_pooled_prompt_embeds: Optional[torch.Tensor] = None,
        guidance_scale: float = 7.5,
        t_next: Optional[torch.Tensor] = None,
        next_latents: Optional[torch.Tensor] = None,
        noise_level: Optional[float] = None,
        joint_attention_kwargs: Optional[Dict[str, Any]] = None,
        compute_log_prob: bool = True,
        return_kwargs: List[str] = ["noise_pred", "next_latents", "next_latents_mean", "std_dev_t", "dt", "log_prob"],
    ) -> SDESchedulerOutput:
        self._before_engine_call()
        self._check_joint_attention_kwargs(joint_attention_kwargs)
        B = latents.shape[0]
        dev = latents.device
        if guidance_scale > 1.0 and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None):
            logger.warning("Passed `guidance_scale` > 1.0, but no `negative_prompt_embeds` or `negative_pooled_prompt_embeds` "
                           "provided. Classifier-free guidance will be disabled.")
        do_cfg = negative_prompt_embeds is not None and negative_pooled_prompt_embeds is not None and guidance_scale > 1.0
        sched = self.scheduler
        t = torch.as_tensor(t, dtype=torch.float32, device=dev)
        if t_next is None:
            idx = [sched.index_for_timestep(x) for x in t.reshape(-1)]
            t_next = torch.stack([sched.timesteps[i + 1] if i + 1 < len(sched.timesteps) else torch.zeros(()) for i in idx]).to(dev)
            if t.ndim == 0:
                t_next = t_next[0]
        t_next = torch.as_tensor(t_next, dtype=torch.float32, device=dev)
        dyn = sched.dynamics_type
        # exact fp32 quotient (torch's GPU tensor/scalar division multiplies by a rounded reciprocal: 1 ulp off
        # the host-side t/1000 the fused rollout uses, which would break the bit-exact ratio == 1 invariant)
        sigma, sigma_next = (t.double() / 1000).float(), (t_next.double() / 1000).float()
        if sched.is_eval or dyn == "ODE":
            noise_level = 0.0
        elif noise_level is None:
            noise_level = sched.get_noise_level_for_sigma(sigma.reshape(-1)) if sigma.ndim else sched.get_noise_level_for_sigma(float(sigma))
        noise = None
        if next_latents is None and dyn != "ODE":
            noise = randn_tensor(latents.shape, device=dev, dtype=torch.float32)  # the draw of flow_match_euler_discrete.py:352-357
        plan = self.engine.plan(B, 2 if do_cfg else 1, latents.shape[2], latents.shape[3], prompt_embeds.shape[1], 1)
        enc = (negative_prompt_embeds, negative_pooled_prompt_embeds, prompt_embeds, pooled_prompt_embeds) if do_cfg else \
              (prompt_embeds, pooled_prompt_embeds, None, None)
        sigma_max = float(sched.sigmas[1])
        want = tuple(k for k in return_kwargs if k in ("next_latents", "next_latents_mean", "noise_pred", "std_dev_t", "dt"))
        view = (-1, 1, 1, 1)
        if torch.is_grad_enabled() and getattr(self, "_live_weights", None) is not None:
            # Grad-mode forward.  (1) `optimize()` replay of a stored transition (trainers/grpo.py:263): log-prob of `next_latents`.
            # (2) The matching-loss trainers -- AWM (trainers/awm.py:357-370), NFT (nft.py:297-304), DPO (dpo.py:468-469), DGPO
            # (dgpo.py:352-364), CRD (crd.py:497-509) -- call forward() WITH autograd at an arbitrary `t`, no stored transition,
            # `compute_log_prob=False`, and read `noise_pred`: the same differentiable engine step, whose network output, mean, std
            # and dt do not depend on the next state (a placeholder transition is passed; its log-prob is switched off).
            # A freshly SAMPLED next state (or its log-prob) with autograd is not a native output: reference path.
            from . import autograd as AG
            replay = next_latents is not None
            sampled = (not replay) and ("next_latents" in return_kwargs or (compute_log_prob and "log_prob" in return_kwargs))
            why = AG.unsupported_reason(self)
            if why is None and sampled:
                why = "a sampled next state (or its log-prob) was requested with autograd"
            if why is None:
                clp = bool(compute_log_prob) and replay
                call = dict(latents=latents, timestep=t, enc_a=enc[0], pooled_a=enc[1], enc_b=enc[2], pooled_b=enc[3], guidance=guidance_scale,
                            sigma=sigma, sigma_next=sigma_next, eta=noise_level, sigma_max=sigma_max, dynamics=dyn,
                            next_latents=next_latents if replay else latents, compute_log_prob=clp)
                lp, npred, mean, std, dtt = AG.denoise_replay(self, plan, call)
                res = dict(noise_pred=npred, next_latents=next_latents.float() if replay else None, next_latents_mean=mean,
                           std_dev_t=std.view(view), dt=dtt.view(view), log_prob=lp if clp else None)
                return self._output_cls.from_dict({k: res[k] for k in return_kwargs if k in res})
            if not why.startswith("the bound module has no trainable"):
                return self._grad_fallback(why, dict(
                    t=t, latents=latents, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                    negative_prompt_embeds=negative_prompt_embeds, negative_pooled_prompt_embeds=negative_pooled_prompt_embeds,
                    guidance_scale=guidance_scale, t_next=t_next, next_latents=next_latents, noise_level=noise_level,
                    joint_attention_kwargs=joint_attention_kwargs, compute_log_prob=compute_log_prob, return_kwargs=return_kwargs))
        o = plan.denoise_step(latents, t, enc[0], enc[1], enc[2], enc[3], guidance_scale, sigma, sigma_next, noise_level,
                              sigma_max, dyn, noise=noise, next_latents=next_latents, compute_log_prob=compute_log_prob, want=want)
        res = dict(
            noise_pred=o.noise_pred,
            next_latents=o.next_latents if next_latents is None else next_latents.float(),
            next_latents_mean=o.next_latents_mean,
            std_dev_t=o.std_dev_t.view(view) if o.std_dev_t is not None else None,
            dt=o.dt.view(view) if o.dt is not None else None,
            log_prob=o.log_prob if compute_log_prob else None,
        )
        return self._output_cls.from_dict({k: res[k] for k in return_kwargs if k in res})


class SD3_5NativeAdapter(NativeRolloutMixin):
    """Standalone adapter (no Flow-Factory import): engine + scheduler + optional VAE decoder."""

    def __init__(self, state_dict: Union[Dict[str, torch.Tensor], torch.nn.Module], config: Optional[TransformerConfig] = None,
                 scheduler: Optional[FlowMatchEulerDiscreteSDEScheduler] = None, latent_storage_dtype: Optional[str] = "fp16",
                 transformer_dtype: torch.dtype = torch.bfloat16, device: Union[str, torch.device] = "cuda",
                 vae_decode: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 vae_state_dict: Optional[Dict[str, torch.Tensor]] = None, vae_config=None, vae_max_batch: int = 4):
        if not torch.cuda.is_available():
            raise RuntimeError("mi355_flow: no GPU visible; the native rollout engine has no CPU path")
        self.device = torch.device(device)
        self.transformer_dtype = transformer_dtype
        self._latent_storage = latent_storage_dtype
        self.scheduler = scheduler or FlowMatchEulerDiscreteSDEScheduler(shift=3.0, sde_steps=[1, 2, 3], num_sde_steps=1)
        self.engine = Engine(config or TransformerConfig())
        self._live_weights = None
        if isinstance(state_dict, torch.nn.Module):
            # a torch module with HF parameter names (possibly DDP / peft wrapped): its CURRENT parameters are re-bound before every
            # engine call, and grad-mode forward() differentiates w.r.t. its trainable parameters (mi355_flow/autograd.py)
            from .binding import LiveWeights
            module = state_dict
            self._live_weights = LiveWeights(self.engine, lambda: module)
            self._sync_weights()
        else:
            self.refresh_weights(state_dict)
        self._vae_decode = vae_decode
        self.vae_decoder = None
        self.vae_max_batch = vae_max_batch
        if vae_state_dict is not None:
            from .vae import VAEConfig, VAEDecoder
            self.vae_decoder = VAEDecoder(vae_config or VAEConfig())
            self.vae_decoder.bind_state_dict(vae_state_dict)
            self.vae_decoder.ready()

    @property
    def latent_storage_dtype(self) -> Optional[torch.dtype]:
        return _DTYPE_MAP.get(self._latent_storage) if self._latent_storage else None

    def _sync_weights(self) -> int:
        if self._live_weights is None:
            return 0
        n = self._live_weights.sync()
        if n:
            self.engine.ready()
        return n

    def _before_engine_call(self) -> None:
        self._sync_weights()

    def refresh_weights(self, state_dict: Dict[str, torch.Tensor]) -> None:
        """Re-pack the (live) torch parameters: call after optimizer steps / EMA swaps / LoRA merges."""
        self.engine.bind_state_dict(state_dict)
        self.engine.ready()

    # mode switches the trainer calls (models/abc.py:351-378)
    def rollout(self):
        self.scheduler.rollout()

    def eval(self):
        self.scheduler.eval()

    def train(self, mode: bool = True):
        self.scheduler.train(mode)

    def encode_prompt(self, *a, **k):
        raise RuntimeError("mi355_flow standalone adapter has no text encoders: pass prompt_embeds / pooled_prompt_embeds "
                           "(the Flow-Factory plugin inherits encode_prompt from SD3_5Adapter)")

    def decode_latents(self, latents: torch.Tensor, output_type: str = "pt"):
        """sd3_5.py:161-172.  With `vae_state_dict` the native decoder (mi355_vae_*) produces the 'pt' images; a custom
        `vae_decode` callable takes precedence; with neither the samples carry no image."""
        if self._vae_decode is not None:
            return self._vae_decode(latents)
        if self.vae_decoder is None:
            return None
        if output_type not in ("pt", "np"):
            raise ValueError("mi355_flow standalone adapter decodes to 'pt' or 'np' (PIL conversion lives in the pipeline's image_processor)")
        img = self.vae_decoder.decode(latents, postprocess=True, out_dtype=torch.bfloat16, max_batch=self.vae_max_batch)
        return img if output_type == "pt" else img.float().permute(0, 2, 3, 1).cpu().numpy()
