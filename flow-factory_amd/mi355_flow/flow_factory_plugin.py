"""The binding a Flow-Factory installation uses:  `model.model_type: mi355_flow.flow_factory_plugin.SD3_5NativeAdapter`.

Flow-Factory resolves an unknown `model_type` as a python path (reference
src/flow_factory/models/registry.py:69-82) and constructs `cls(config=config, accelerator=accelerator)`
(models/loader.py:61-64).  The class below IS the reference's `SD3_5Adapter` (pipeline loading, text
encoders, VAE, LoRA, EMA, checkpoints, device placement all inherited) with the rollout hot path
(`inference`, no-grad `forward`) routed to libmi355flow.so through `NativeRolloutMixin`.

Importing this module needs `flow_factory` (+ diffusers, peft): it is NOT importable in the build
container; `mi355_flow.adapter` carries the same code path standalone and is what the tests run.
"""
from __future__ import annotations

import torch

try:  # pragma: no cover - exercised only inside a Flow-Factory installation
    from flow_factory.models.stable_diffusion.sd3_5 import SD3_5Adapter as _RefAdapter
except Exception as e:  # noqa: BLE001
    _RefAdapter = None
    _IMPORT_ERROR = e

from .adapter import NativeRolloutMixin
from .engine import Engine, TransformerConfig
from .vae import VAEConfig, VAEDecoder

if _RefAdapter is not None:  # pragma: no cover

    class SD3_5NativeAdapter(NativeRolloutMixin, _RefAdapter):
        """`SD3_5Adapter` with the GRPO rollout running on the MI355X engine."""

        def __init__(self, config, accelerator):
            _RefAdapter.__init__(self, config, accelerator)
            tc = self.pipeline.transformer.config
            self.engine = Engine(TransformerConfig(
                in_channels=tc.in_channels, out_channels=tc.out_channels, patch_size=tc.patch_size,
                num_layers=tc.num_layers, num_heads=tc.num_attention_heads, head_dim=tc.attention_head_dim,
                joint_attention_dim=tc.joint_attention_dim, pooled_projection_dim=tc.pooled_projection_dim,
                pos_embed_max_size=tc.pos_embed_max_size, dual_layers=tuple(tc.dual_attention_layers)))
            self._bound_version = -1
            self._weights_version = 0
            # the VAE is frozen (`_freeze_vae`, models/abc.py): bind its decoder once
            self.vae_decoder = VAEDecoder(VAEConfig.from_hf(self.pipeline.vae.config))
            self.vae_decoder.bind_state_dict(self.pipeline.vae.state_dict())
            self.vae_decoder.ready()

        # the engine computes in bf16 like the reference's autocast run
        @property
        def transformer_dtype(self):
            return self.pipeline.transformer.dtype

        def _sync_weights(self):
            if self._bound_version != self._weights_version:
                module = self.accelerator.unwrap_model(self.transformer)
                self.engine.bind_state_dict(module.state_dict())  # LoRA: merge first (peft `merge_adapter`) or bind merged weights
                self.engine.ready()
                self._bound_version = self._weights_version

        # weights are live: every mode switch that can change them invalidates the packed copy
        def rollout(self, *a, **k):
            self._weights_version += 1
            return _RefAdapter.rollout(self, *a, **k)

        def eval(self, *a, **k):
            self._weights_version += 1
            return _RefAdapter.eval(self, *a, **k)

        @torch.no_grad()
        def decode_latents(self, latents, output_type="pil"):
            # sd3_5.py:161-172 with vae.decode on the native decoder; 'pt' is what the rollout asks for (sd3_5.py:307)
            if output_type == "pt":
                return self.vae_decoder.decode(latents, postprocess=True, out_dtype=torch.bfloat16)
            images = self.vae_decoder.decode(latents, postprocess=False, out_dtype=torch.bfloat16)
            return self.pipeline.image_processor.postprocess(images, output_type=output_type)

        @torch.no_grad()
        def inference(self, *args, **kwargs):
            self._sync_weights()
            return NativeRolloutMixin.inference(self, *args, **kwargs)

        def forward(self, *args, **kwargs):
            # optimize() (trainers/grpo.py:263) needs autograd through the transformer: that stays on the
            # reference path until the backward kernels exist.  The rollout is no-grad.
            if torch.is_grad_enabled():
                return _RefAdapter.forward(self, *args, **kwargs)
            self._sync_weights()
            return NativeRolloutMixin.forward(self, *args, **kwargs)

    try:
        from flow_factory.models.flux.flux1 import Flux1Adapter as _RefFlux
    except Exception:  # noqa: BLE001
        _RefFlux = None

    if _RefFlux is not None:
        from .flux import Flux1NativeAdapter as _FluxMirror, FluxConfig, FluxEngine, unpack_latents

        class Flux1NativeAdapter(_RefFlux):
            """`Flux1Adapter` (flux1.py) with the GRPO rollout on the MI355X engine: `model.model_type:
            mi355_flow.flow_factory_plugin.Flux1NativeAdapter`.  inference() / no-grad forward() / decode_latents() are the
            standalone mirror's methods (same signatures as the reference's), bound onto the reference adapter."""

            def __init__(self, config, accelerator):
                _RefFlux.__init__(self, config, accelerator)
                tc = self.pipeline.transformer.config
                self.engine = FluxEngine(FluxConfig(
                    in_channels=tc.in_channels, num_layers=tc.num_layers, num_single_layers=tc.num_single_layers,
                    num_attention_heads=tc.num_attention_heads, attention_head_dim=tc.attention_head_dim,
                    joint_attention_dim=tc.joint_attention_dim, pooled_projection_dim=tc.pooled_projection_dim,
                    guidance_embeds=bool(tc.guidance_embeds), axes_dims_rope=tuple(tc.axes_dims_rope)))
                self._bound_version, self._weights_version = -1, 0
                self.vae_max_batch = 4
                self.vae_decoder = VAEDecoder(VAEConfig.from_hf(self.pipeline.vae.config))
                self.vae_decoder.bind_state_dict(self.pipeline.vae.state_dict())
                self.vae_decoder.ready()

            @property
            def transformer_dtype(self):
                return self.pipeline.transformer.dtype

            def _sync_weights(self):
                if self._bound_version != self._weights_version:
                    self.engine.bind_state_dict(self.accelerator.unwrap_model(self.transformer).state_dict())
                    self.engine.ready()
                    self._bound_version = self._weights_version

            def rollout(self, *a, **k):
                self._weights_version += 1
                return _RefFlux.rollout(self, *a, **k)

            def eval(self, *a, **k):
                self._weights_version += 1
                return _RefFlux.eval(self, *a, **k)

            @torch.no_grad()
            def decode_latents(self, latents, height, width, output_type="pil"):
                lat = unpack_latents(latents, int(height) // 8, int(width) // 8).contiguous()
                if output_type == "pt":
                    return self.vae_decoder.decode(lat, postprocess=True, out_dtype=torch.bfloat16, max_batch=self.vae_max_batch)
                images = self.vae_decoder.decode(lat, postprocess=False, out_dtype=torch.bfloat16, max_batch=self.vae_max_batch)
                return self.pipeline.image_processor.postprocess(images, output_type=output_type)

            @torch.no_grad()
            def inference(self, *args, **kwargs):
                self._sync_weights()
                return _FluxMirror.inference(self, *args, **kwargs)

            def forward(self, *args, **kwargs):
                if torch.is_grad_enabled():          # optimize(): autograd stays on the reference path
                    return _RefFlux.forward(self, *args, **kwargs)
                self._sync_weights()
                return _FluxMirror.forward(self, *args, **kwargs)

    try:
        from flow_factory.models.wan.wan2_t2v import Wan2_T2V_Adapter as _RefWan
    except Exception:  # noqa: BLE001
        _RefWan = None

    if _RefWan is not None:
        from .wan import Wan2T2VNativeAdapter as _WanMirror, WanConfig, WanEngine

        class Wan2T2VNativeAdapter(_RefWan):
            """`Wan2_T2V_Adapter` (wan2_t2v.py) with the Wan2.1 rollout on the MI355X engine (single transformer; Wan2.2's
            `transformer_2` / `boundary_ratio` configurations stay on the reference path).  The video VAE is the pipeline's."""

            def __init__(self, config, accelerator):
                _RefWan.__init__(self, config, accelerator)
                if getattr(self.pipeline.config, "boundary_ratio", None) is not None or getattr(self.pipeline, "transformer_2", None) is not None:
                    raise ValueError("mi355_flow: two-expert Wan2.2 pipelines are not supported by the native engine")
                tc = self.pipeline.transformer.config
                self.engine = WanEngine(WanConfig(
                    in_channels=tc.in_channels, out_channels=tc.out_channels, num_layers=tc.num_layers,
                    num_attention_heads=tc.num_attention_heads, attention_head_dim=tc.attention_head_dim, ffn_dim=tc.ffn_dim,
                    text_dim=tc.text_dim, freq_dim=tc.freq_dim, patch_size=tuple(tc.patch_size), eps=tc.eps))
                self._bound_version, self._weights_version = -1, 0

            @property
            def transformer_dtype(self):
                return self.pipeline.transformer.dtype

            def _sync_weights(self):
                if self._bound_version != self._weights_version:
                    self.engine.bind_state_dict(self.accelerator.unwrap_model(self.transformer).state_dict())
                    self.engine.ready()
                    self._bound_version = self._weights_version

            def rollout(self, *a, **k):
                self._weights_version += 1
                return _RefWan.rollout(self, *a, **k)

            def eval(self, *a, **k):
                self._weights_version += 1
                return _RefWan.eval(self, *a, **k)

            @torch.no_grad()
            def inference(self, *args, **kwargs):
                self._sync_weights()
                return _WanMirror.inference(self, *args, **kwargs)

            def forward(self, *args, **kwargs):
                if torch.is_grad_enabled():
                    return _RefWan.forward(self, *args, **kwargs)
                self._sync_weights()
                return _WanMirror.forward(self, *args, **kwargs)

else:

    class SD3_5NativeAdapter:  # type: ignore[no-redef]
        def __init__(self, *a, **k):
            raise ImportError("mi355_flow.flow_factory_plugin needs an importable `flow_factory` (with diffusers/peft): "
                              f"{_IMPORT_ERROR!r}.  Use mi355_flow.adapter.SD3_5NativeAdapter standalone instead.")

    class Wan2T2VNativeAdapter:  # type: ignore[no-redef]
        def __init__(self, *a, **k):
            raise ImportError("mi355_flow.flow_factory_plugin needs an importable `flow_factory` (with diffusers/peft): "
                              f"{_IMPORT_ERROR!r}.  Use mi355_flow.wan.Wan2T2VNativeAdapter standalone instead.")

    class Flux1NativeAdapter:  # type: ignore[no-redef]
        def __init__(self, *a, **k):
            raise ImportError("mi355_flow.flow_factory_plugin needs an importable `flow_factory` (with diffusers/peft): "
                              f"{_IMPORT_ERROR!r}.  Use mi355_flow.flux.Flux1NativeAdapter standalone instead.")
