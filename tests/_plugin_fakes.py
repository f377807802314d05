"""Test doubles for driving `mi355_flow.flow_factory_plugin` on a CPU-only box: a pseudo-pipeline with HF-named torch parameters,
a single-process accelerator and an engine that records what was bound / launched instead of calling libmi355flow.so.
(The real engine needs a GPU; the plugin's Python binding -- class construction, weight liveness, sample classes, kwargs
filtering, mode switches -- does not.)"""
from __future__ import annotations

import types
from typing import Dict, List, Tuple

import torch
import torch.nn as nn


# --------------------------------------------------------------------------------------- modules with HF parameter names
def build_module_tree(shapes: Dict[str, Tuple[int, ...]], buffers=("pos_embed.pos_embed",), seed=0, std=0.05, cls=nn.Module) -> nn.Module:
    """Nested nn.Modules whose `named_parameters()` / attribute paths are exactly `shapes`' keys."""
    g = torch.Generator().manual_seed(seed)
    root = cls()
    for name, shape in shapes.items():
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        val = torch.randn(shape, generator=g) * std
        if ".norm_" in name and len(shape) == 1:
            val = val + 1.0
        if name in buffers:
            mod.register_buffer(parts[-1], val)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(val))
    return root


class FakeTransformer(nn.Module):
    """Parameters only: the native path must never call the torch forward."""

    calls = 0

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def forward(self, *a, **k):
        type(self).calls += 1
        raise AssertionError("the torch transformer forward was called on the native rollout path")


class FakeLoraLinear(nn.Module):
    """Shape of a peft `lora.Linear`: base_layer + lora_A / lora_B ModuleDicts + scaling, adapters can be disabled."""

    def __init__(self, base: nn.Module, r=4, alpha=8.0, seed=0):
        super().__init__()
        out_f, in_f = base.weight.shape
        self.base_layer = base
        g = torch.Generator().manual_seed(seed)
        a, b = nn.Linear(in_f, r, bias=False), nn.Linear(r, out_f, bias=False)
        with torch.no_grad():
            a.weight.copy_(torch.randn(r, in_f, generator=g) * 0.1)
            b.weight.copy_(torch.randn(out_f, r, generator=g) * 0.1)
        self.lora_A, self.lora_B = nn.ModuleDict({"default": a}), nn.ModuleDict({"default": b})
        self.scaling = {"default": alpha / r}
        self.active_adapters = ["default"]
        self.disable_adapters = False
        self.merged = False
        base.weight.requires_grad_(False)


def wrap_lora(root: nn.Module, targets=("to_q", "to_k", "to_v", "to_out.0")) -> List[str]:
    """Replace every `...<target>` leaf (a module holding weight/bias) by a FakeLoraLinear; returns the wrapped paths."""
    wrapped = []
    for path, mod in list(root.named_modules()):
        if any(path.endswith("." + t) for t in targets) and hasattr(mod, "weight"):
            parent = root.get_submodule(path.rsplit(".", 1)[0])
            setattr(parent, path.rsplit(".", 1)[1], FakeLoraLinear(mod, seed=len(wrapped)))
            wrapped.append(path)
    return wrapped


class FakePeftModel(nn.Module):
    """`PeftModel(base_model=LoraModel(model=<transformer>))` nesting with a `disable_adapter()` context."""

    def __init__(self, inner: nn.Module):
        super().__init__()
        self.peft_config = {"default": object()}
        self.base_model = nn.Module()
        self.base_model.add_module("model", inner)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def disable_adapter(self):
        from contextlib import contextmanager

        @contextmanager
        def ctx():
            layers = [m for m in self.modules() if isinstance(m, FakeLoraLinear)]
            for m in layers:
                m.disable_adapters = True
            try:
                yield
            finally:
                for m in layers:
                    m.disable_adapters = False
        return ctx()


# --------------------------------------------------------------------------------------- engine double
class FakePlan:
    def __init__(self, engine, key, max_steps):
        self.engine, self.key, self.max_steps = engine, key, max_steps
        self.batch, self.n_cfg, self.h, self.w, self.n_text = key
        self.C = engine.cfg.in_channels

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance, init_latents, storage_dtype, step_noise, prompt_embeds, pooled,
                neg_embeds=None, neg_pooled=None, keep_positions=None, compute_log_prob=True):
        N, B = len(timesteps), self.batch
        self.engine.calls.append(("rollout", dict(N=N, dynamics=dynamics, guidance=guidance, noise_levels=list(noise_levels),
                                                   keep=list(keep_positions) if keep_positions is not None else None,
                                                   weights=self.engine.fingerprint())))
        keep = list(range(N + 1)) if keep_positions is None else sorted(set(keep_positions))
        shape = (B, self.C, self.h, self.w)
        g = torch.Generator().manual_seed(5)
        lat = torch.randn((len(keep),) + shape, generator=g).to(storage_dtype)
        fin = lat[-1].clone() if keep else torch.randn(shape, generator=g).to(storage_dtype)      # (nothing kept: evaluation)
        lp = torch.full((N, B), float("nan"))
        for i, e in enumerate(noise_levels):
            if e > 0 and compute_log_prob:
                lp[i] = -1.0 - 0.01 * i - 0.001 * torch.arange(B)
        return lat, lp, fin

    def denoise_step(self, latents, timestep, enc_a, pooled_a, enc_b, pooled_b, guidance, sigma, sigma_next, eta, sigma_max, dynamics,
                     noise=None, next_latents=None, compute_log_prob=True, want=()):
        B = latents.shape[0]
        self.engine.calls.append(("denoise_step", dict(replay=next_latents is not None, weights=self.engine.fingerprint(), eta=eta)))
        o = types.SimpleNamespace()
        o.next_storage = latents.clone()
        for k in ("next_latents", "next_latents_mean", "noise_pred"):
            setattr(o, k, latents.float().clone() if k in want else None)
        o.log_prob = torch.full((B,), -1.0) if compute_log_prob else None
        o.std_dev_t = torch.full((B,), 0.5) if "std_dev_t" in want else None
        o.dt = torch.full((B,), -0.1) if "dt" in want else None
        return o


class FakeEngine:
    _ABI, _WHAT = "engine", "transformer"
    """Stands in for mi355_flow.engine.Engine: same Python surface, keeps the bound tensors (fp32 copies) for inspection."""

    def __init__(self, cfg):
        from mi355_flow.weights import expected_shapes
        self.cfg = cfg
        self._names = list(expected_shapes(cfg).keys())
        self.bound: Dict[str, torch.Tensor] = {}
        self.bind_log: List[str] = []
        self.calls: List[tuple] = []
        self._plans = {}

    def param_names(self):
        return self._names

    def bind_tensor(self, name, t):
        assert name in self._names, name
        self.bound[name] = t.detach().float().clone()
        self.bind_log.append(name)

    def finish_binding(self):
        pass

    def bind_state_dict(self, sd, partial=False, strict=None):
        from mi355_flow.engine import WeightHolder
        WeightHolder.bind_state_dict(self, sd, partial=partial, strict=strict)

    def ready(self):
        missing = [n for n in self._names if n not in self.bound]
        if missing:
            raise RuntimeError(f"parameter '{missing[0]}' has not been bound")

    def fingerprint(self) -> float:
        return float(sum(float(v.double().sum()) for v in self.bound.values()))

    def plan(self, batch, n_cfg, h, w, n_text, max_steps):
        key = (batch, n_cfg, h, w, n_text)
        p = self._plans.get(key)
        if p is None or p.max_steps < max_steps:
            p = self._plans[key] = FakePlan(self, key, max_steps)
        return p

    def close(self):
        pass


class FakeVAEDecoder:
    def __init__(self, cfg=None):
        self.n = 0

    def bind_state_dict(self, sd, **k):
        pass

    def ready(self):
        pass

    def decode(self, latents, postprocess=True, out_dtype=torch.bfloat16, max_batch=4):
        self.n += 1
        B, _, h, w = latents.shape
        return torch.full((B, 3, 8 * h, 8 * w), 0.5, dtype=out_dtype)


# --------------------------------------------------------------------------------------- pipeline / accelerator doubles
class FakeAccelerator:
    def __init__(self):
        self.device = torch.device("cpu")
        self.is_main_process = True
        self.process_index, self.num_processes = 0, 1
        self.sync_gradients = True
        self.state = types.SimpleNamespace(deepspeed_plugin=None, fsdp_plugin=None)
        self.distributed_type = "NO"

    def unwrap_model(self, m, **k):
        return m

    def prepare(self, *mods):
        return mods if len(mods) != 1 else mods[0]


def make_pipeline(cfg, transformer: nn.Module):
    """A "pseudo-pipeline" (reference guidance/new_model.md:574-718): a plain object exposing flat component attributes."""
    from oracle import diffusers_stub as D

    tc = types.SimpleNamespace(
        in_channels=cfg.in_channels, out_channels=cfg.out_channels, patch_size=cfg.patch_size, num_layers=cfg.num_layers,
        num_attention_heads=cfg.num_heads, attention_head_dim=cfg.head_dim, joint_attention_dim=cfg.joint_attention_dim,
        pooled_projection_dim=cfg.pooled_projection_dim, pos_embed_max_size=cfg.pos_embed_max_size,
        dual_attention_layers=tuple(cfg.dual_layers))
    transformer.config = tc
    vae = nn.Module()
    vae.add_module("decoder", nn.Linear(2, 2))
    vae.config = types.SimpleNamespace(scaling_factor=1.5305, shift_factor=0.0609, latent_channels=16, block_out_channels=(128, 256, 512, 512),
                                       layers_per_block=2, norm_num_groups=32, out_channels=3)
    pipe = types.SimpleNamespace()
    pipe.transformer = transformer
    pipe.vae = vae
    pipe.text_encoder = nn.Linear(2, 2)
    pipe.tokenizer = object()
    pipe.tokenizer_3 = object()
    pipe.scheduler = D.FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=False)
    pipe.image_processor = types.SimpleNamespace(postprocess=lambda x, output_type="pt": x)
    pipe.maybe_free_model_hooks = lambda: None
    pipe.components = {"transformer": transformer, "vae": vae, "text_encoder": pipe.text_encoder}
    return pipe


# --------------------------------------------------------------------------------------- Qwen-Image doubles
class FakeQwenPlan:
    def __init__(self, engine, key, max_steps):
        self.engine, self.key, self.max_steps = engine, key, max_steps
        self.batch, self.n_cfg, self.h, self.w, self.n_text = key
        self.Ni, self.C = (self.h // 2) * (self.w // 2), engine.cfg.in_channels

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance, init_latents, storage_dtype, step_noise, embeds, lens=None,
                keep_positions=None, compute_log_prob=True):
        N, B = len(timesteps), self.batch
        assert embeds.shape[0] == B * self.n_cfg and embeds.shape[1] == self.n_text and len(lens) == B * self.n_cfg
        self.engine.calls.append(("rollout", dict(N=N, dynamics=dynamics, guidance=guidance, noise_levels=list(noise_levels), n_cfg=self.n_cfg,
                                                   lens=list(lens), n_text=self.n_text, keep=list(keep_positions) if keep_positions is not None else None,
                                                   weights=self.engine.fingerprint())))
        keep = list(range(N + 1)) if keep_positions is None else sorted(set(keep_positions))
        g = torch.Generator().manual_seed(5)
        lat = torch.randn((len(keep), B, self.Ni, self.C), generator=g).to(storage_dtype)
        fin = lat[-1].clone() if keep else torch.randn((B, self.Ni, self.C), generator=g).to(storage_dtype)
        lp = torch.full((N, B), float("nan"))
        for i, e in enumerate(noise_levels):
            if e > 0 and compute_log_prob:
                lp[i] = -1.0 - 0.01 * i - 0.001 * torch.arange(B)
        return lat, lp, fin


class FakeQwenEngine(FakeEngine):
    _ABI, _WHAT = "qwen", "Qwen-Image transformer"

    def __init__(self, cfg):
        from oracle import qwen_ref as Q
        self.cfg = cfg
        self._names = list(Q.state_dict_shapes(Q.QwenConfig(num_layers=cfg.num_layers, num_attention_heads=cfg.num_attention_heads,
                                                            joint_attention_dim=cfg.joint_attention_dim)).keys())
        self.bound, self.bind_log, self.calls, self._plans = {}, [], [], {}

    def plan(self, batch, n_cfg, h, w, n_text, max_steps):
        key = (batch, n_cfg, h, w, n_text)
        p = self._plans.get(key)
        if p is None or p.max_steps < max_steps:
            p = self._plans[key] = FakeQwenPlan(self, key, max_steps)
        return p


class FakeVideoVAEDecoder:
    def __init__(self, cfg=None):
        self.n = 0

    def bind_state_dict(self, sd, **k):
        pass

    def ready(self):
        pass

    def decode(self, latents, postprocess=True, out_dtype=torch.bfloat16, max_batch=1, denormalise=True):
        self.n += 1
        B, _, T, h, w = latents.shape
        return torch.full((B, 1 + 4 * (T - 1), 3, 8 * h, 8 * w), 0.5, dtype=out_dtype)


def make_qwen_pipeline(tcfg, transformer: nn.Module):
    from oracle import diffusers_stub as D
    transformer.config = types.SimpleNamespace(
        in_channels=tcfg.in_channels, num_layers=tcfg.num_layers, num_attention_heads=tcfg.num_attention_heads,
        attention_head_dim=tcfg.attention_head_dim, joint_attention_dim=tcfg.joint_attention_dim, axes_dims_rope=tuple(tcfg.axes_dims_rope))
    vae = nn.Module()
    vae.add_module("decoder", nn.Linear(2, 2))
    vae.config = types.SimpleNamespace(z_dim=16, base_dim=96, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True),
                                       latents_mean=[0.0] * 16, latents_std=[1.0] * 16)
    pipe = types.SimpleNamespace()
    pipe.transformer, pipe.vae = transformer, vae
    pipe.text_encoder = nn.Linear(2, 2)
    pipe.tokenizer = object()
    pipe.vae_scale_factor = 8
    pipe.scheduler = D.FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=True, base_shift=0.5,
                                                       max_shift=0.9, base_image_seq_len=256, max_image_seq_len=8192)
    pipe.image_processor = types.SimpleNamespace(postprocess=lambda x, output_type="pt": x)
    pipe.maybe_free_model_hooks = lambda: None
    pipe.components = {"transformer": transformer, "vae": vae, "text_encoder": pipe.text_encoder}
    return pipe


# --------------------------------------------------------------------------------------- differentiable engine double
class DiffFakePlan(FakePlan):
    """FakePlan whose log-prob is one fixed function of (timestep, sample index, the weights bound NOW) in the rollout, the no-grad step and
    the training step alike -- so `ratio == 1` holds before an update and moves after one -- with the native training-step API of
    mi355_flow.engine.Plan (`denoise_step_train` / `denoise_step_backward`) that mi355_flow.autograd drives."""

    C_W = 50.0          # large enough that an SGD step on the (1 / numel)-sized gradients survives fp32 rounding

    def _lp(self, t, latents):
        # a function of the timestep, the sample's OWN current latents and the weights bound now (no dependence on the position inside
        # the micro-batch: optimize() shuffles the samples into other micro-batches)
        B = latents.shape[0]
        t = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
        t = t.expand(B) if t.numel() == 1 else t[:B]
        m = latents.float().reshape(B, -1).mean(1)
        return (-1.0 - 1e-4 * t + self.C_W * self.engine.weight_signal() * (1.0 + m)).float()

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance, init_latents, storage_dtype, step_noise, prompt_embeds, pooled,
                neg_embeds=None, neg_pooled=None, keep_positions=None, compute_log_prob=True):
        lat, lp, fin = super().rollout(timesteps, sigmas, noise_levels, dynamics, guidance, init_latents, storage_dtype, step_noise,
                                       prompt_embeds, pooled, neg_embeds, neg_pooled, keep_positions, compute_log_prob)
        N = len(timesteps)
        keep = list(range(N + 1)) if keep_positions is None else sorted(set(keep_positions))
        for i, e in enumerate(noise_levels):
            if e > 0 and compute_log_prob:
                lp[i] = self._lp(timesteps[i], lat[keep.index(i)])
        return lat, lp, fin

    def denoise_step(self, latents, timestep, *a, **k):
        o = super().denoise_step(latents, timestep, *a, **k)
        if o.log_prob is not None:
            o.log_prob = self._lp(timestep, latents)
        if o.noise_pred is not None:
            o.noise_pred = latents.float() * (1.0 + self.engine.weight_signal())
        return o

    def denoise_step_train(self, latents, timestep, enc_a, pooled_a, enc_b, pooled_b, guidance, sigma, sigma_next, eta, sigma_max, dynamics,
                           next_latents, compute_log_prob=True, _keep=None):
        B = latents.shape[0]
        self.engine.calls.append(("denoise_step_train", dict(weights=self.engine.fingerprint(), eta=eta, clp=bool(compute_log_prob))))
        o = types.SimpleNamespace()
        o.log_prob = self._lp(timestep, latents)
        o.noise_pred = latents.float() * (1.0 + self.engine.weight_signal())
        o.next_latents_mean = latents.float().clone()
        o.std_dev_t, o.dt = torch.full((B,), 0.5), torch.full((B,), -0.1)
        if _keep is not None:
            _keep.update(latents=latents, B=B)
        return o

    def denoise_step_backward(self, call, g_log_prob, g_noise_pred, g_mean):
        # d lp_b / d w = C_W / numel(w) for every element of every signal tensor; d noise_pred / d w = latents / numel(w)
        eng, lat = self.engine, call["_keep"]["latents"].float()
        self.engine.calls.append(("denoise_step_backward", dict(has_lp=g_log_prob is not None, has_np=g_noise_pred is not None)))
        up = 0.0
        if g_log_prob is not None:
            up += self.C_W * float((g_log_prob.float() * (1.0 + lat.reshape(lat.shape[0], -1).mean(1))).sum())
        if g_noise_pred is not None:
            up += float((g_noise_pred.float() * lat).sum())
        for name, buf in eng.grad_bufs.items():
            if name in eng.signal_names():
                buf += up / (buf.numel() * len(eng.signal_names()))


class DiffFakeEngine(FakeEngine):
    """FakeEngine + the gradient-registration surface of mi355_flow.engine.Engine.  `weight_signal()` = mean over the attention-projection
    weights of block 0 of what is bound now (they are in the reference's default `target_modules`, so an optimizer step moves it)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.grad_bufs: Dict[str, torch.Tensor] = {}
        self.scope_full = None

    def signal_names(self):
        return [n for n in self._names if n.startswith("transformer_blocks.0.attn.to_") and n.endswith(".weight")]

    def weight_signal(self) -> float:
        names = self.signal_names()
        return float(sum(float(self.bound[n].double().mean()) for n in names) / len(names))

    def plan(self, batch, n_cfg, h, w, n_text, max_steps):
        key = (batch, n_cfg, h, w, n_text)
        p = self._plans.get(key)
        if p is None or p.max_steps < max_steps:
            p = self._plans[key] = DiffFakePlan(self, key, max_steps)
        return p

    def grad_supported(self, name) -> int:
        return 1 if name in self._names else 0

    def set_train_scope(self, full) -> None:
        self.scope_full = bool(full)

    def set_grad(self, name, t) -> None:
        assert name in self._names and t.dtype in (torch.float32, torch.bfloat16)
        self.grad_bufs[name] = t

    def clear_grads(self) -> None:
        self.grad_bufs = {}


class TrainerAccelerator(FakeAccelerator):
    """Single-process stand-in for what the reference's trainers / AdvantageProcessor / reduce_loss_info ask of `accelerate.Accelerator`."""

    mixed_precision = "bf16"
    is_local_main_process = True

    def accumulate(self, *models):
        from contextlib import nullcontext
        return nullcontext()

    def backward(self, loss):
        loss.backward()

    def clip_grad_norm_(self, params, max_norm):
        return torch.nn.utils.clip_grad_norm_(list(params), max_norm)

    def gather(self, t):
        return t

    def reduce(self, t, reduction="mean"):
        return t

    def wait_for_everyone(self):
        pass


class DistTrainerAccelerator(TrainerAccelerator):
    """`TrainerAccelerator` over an initialised `torch.distributed` group (gloo on CPU): what accelerate's gather / reduce / barrier do on
    N processes, for world-size-2 runs of the reference's trainers through the plugin (tests/test_dist_gloo.py)."""

    def __init__(self):
        import torch.distributed as dist
        super().__init__()
        self.process_index, self.num_processes = dist.get_rank(), dist.get_world_size()
        self.is_main_process = self.is_local_main_process = self.process_index == 0
        self.distributed_type = "MULTI_CPU"

    def gather(self, t):
        import torch.distributed as dist
        t = t.contiguous()
        out = [torch.empty_like(t) for _ in range(self.num_processes)]
        dist.all_gather(out, t)
        return torch.cat(out, 0)

    def reduce(self, t, reduction="mean"):
        import torch.distributed as dist
        if isinstance(t, dict):                                   # accelerate reduces nested structures leaf by leaf
            return {k: self.reduce(v, reduction) for k, v in t.items()}
        if isinstance(t, (list, tuple)):
            return type(t)(self.reduce(v, reduction) for v in t)
        t = t.clone()
        dist.all_reduce(t)
        return t / self.num_processes if reduction == "mean" else t

    def wait_for_everyone(self):
        import torch.distributed as dist
        dist.barrier()

    def unwrap_model(self, m, **k):
        return m.module if type(m).__name__ == "DistributedDataParallel" else m


# --------------------------------------------------------------------------------------- engine double that computes (oracle + stand-in)
class StandinPlan(FakePlan):
    """Plan double whose rollout / single step ARE computed: the oracle's SDE step (`oracle.scheduler_ref`, pinned bit-exact against the
    reference's scheduler) around `oracle.standin.denoiser` in place of the network.  With it the plugin's host path -- RNG draws, schedule,
    noise levels, kept positions, collectors, sample construction -- can be compared with the reference adapter's results on CPU."""

    def _net(self, latents, t, enc_a, pool_a, enc_b, pool_b, guidance):
        from oracle import rollout_ref as R
        from oracle import standin
        B = latents.shape[0]
        tt = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
        timestep = (tt.expand(B) if tt.numel() == 1 else tt).to(latents.dtype)
        if enc_b is not None:                              # CFG: (enc_a, pool_a) = negative, (enc_b, pool_b) = positive; batch order [neg, pos]
            v = standin.denoiser(torch.cat([latents, latents]), timestep.repeat(2), torch.cat([enc_a, enc_b]), torch.cat([pool_a, pool_b]))
            vu, vt = v.chunk(2)
            return R.cfg_combine_bf16(vu, vt, guidance)
        return standin.denoiser(latents, timestep, enc_a, pool_a)

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance, init_latents, storage_dtype, step_noise, prompt_embeds, pooled,
                neg_embeds=None, neg_pooled=None, keep_positions=None, compute_log_prob=True):
        from oracle import rollout_ref as R
        from oracle import standin
        N = len(timesteps)
        self.engine.calls.append(("rollout", dict(N=N, dynamics=dynamics, guidance=guidance, noise_levels=list(noise_levels),
                                                   keep=list(keep_positions) if keep_positions is not None else None,
                                                   weights=self.engine.fingerprint())))
        if step_noise is None:
            step_noise = torch.zeros((N,) + tuple(init_latents.shape))
        out = R.rollout(None, None, prompt_embeds, pooled, neg_embeds, neg_pooled, guidance, init_latents, step_noise,
                        torch.tensor(timesteps, dtype=torch.float32), torch.tensor(sigmas, dtype=torch.float32), list(noise_levels), storage_dtype,
                        dynamics_type=dynamics, compute_log_prob=compute_log_prob, denoiser=standin.denoiser)
        keep = list(range(N + 1)) if keep_positions is None else sorted(set(keep_positions))
        kept = torch.stack([out["all_latents"][p] for p in keep]) if keep else out["all_latents"][:0]
        return kept, out["log_probs"], out["all_latents"][N]

    def denoise_step(self, latents, timestep, enc_a, pooled_a, enc_b, pooled_b, guidance, sigma, sigma_next, eta, sigma_max, dynamics,
                     noise=None, next_latents=None, compute_log_prob=True, want=()):
        from oracle import scheduler_ref as S
        self.engine.calls.append(("denoise_step", dict(replay=next_latents is not None, weights=self.engine.fingerprint(), eta=eta)))
        v = self._net(latents, timestep, enc_a, pooled_a, enc_b, pooled_b, guidance).to(torch.bfloat16)
        if torch.is_tensor(eta):                          # a per-sample noise level inferred from per-sample sigmas: one value in practice
            assert bool((eta == eta.reshape(-1)[0]).all()), "the double serves one noise level per call"
            eta = eta.reshape(-1)[0]
        out = S.sde_step(v, latents, torch.as_tensor(sigma, dtype=torch.float32), torch.as_tensor(sigma_next, dtype=torch.float32), float(eta),
                         dynamics_type=dynamics, sigma_max=sigma_max, variance_noise=noise, next_latents=next_latents,
                         compute_log_prob=compute_log_prob)
        o = types.SimpleNamespace(**{k: out.get(k) for k in ("next_latents", "next_latents_mean", "noise_pred", "log_prob", "std_dev_t", "dt")})
        o.next_storage = S.cast_latents(out["next_latents"], latents.dtype)
        return o


class StandinEngine(FakeEngine):
    def plan(self, batch, n_cfg, h, w, n_text, max_steps):
        key = (batch, n_cfg, h, w, n_text)
        p = self._plans.get(key)
        if p is None or p.max_steps < max_steps:
            p = self._plans[key] = StandinPlan(self, key, max_steps)
        return p


class _StandinFamilyEngine(FakeEngine):
    """Computing engine double for the FLUX.1 / Qwen-Image / Wan plugin classes: parameter names = whatever tiny module tree the test
    binds (`NAMES`), `plan(...).rollout(...)` = the family's ORACLE rollout around its stand-in network."""

    NAMES: List[str] = []
    PLAN = None

    def __init__(self, cfg):
        self.cfg = cfg
        self._names = list(type(self).NAMES)
        self.bound, self.bind_log, self.calls, self._plans = {}, [], [], {}

    def plan(self, *key):
        p = self._plans.get(key)
        if p is None:
            p = self._plans[key] = type(self).PLAN(self, key)
        return p


def _keep_rows(out, N, keep_positions):
    keep = list(range(N + 1)) if keep_positions is None else sorted(set(keep_positions))
    kept = torch.stack([out["all_latents"][p] for p in keep]) if keep else out["all_latents"][:0]
    return kept, out["log_probs"], out["all_latents"][N]


class FluxStandinPlan:
    def __init__(self, engine, key):
        self.engine = engine
        self.batch, self.h, self.w, self.n_text, self.max_steps = key

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance_scale, init_latents, storage_dtype, step_noise, prompt_embeds, pooled,
                keep_positions=None, compute_log_prob=True):
        from oracle import flux_ref as FR
        from oracle import standin
        self.engine.calls.append(("rollout", dict(N=len(timesteps))))
        if step_noise is None:                               # ODE dynamics: the plugin draws (and the step uses) no noise
            step_noise = torch.zeros((len(timesteps),) + tuple(init_latents.shape))
        out = FR.rollout(None, None, prompt_embeds, pooled, guidance_scale, init_latents, step_noise, torch.tensor(timesteps, dtype=torch.float32),
                         torch.tensor(sigmas, dtype=torch.float32), list(noise_levels),
                         FR.prepare_img_ids(self.h // 2, self.w // 2).to(init_latents.dtype), storage_dtype, dynamics_type=dynamics,
                         compute_log_prob=compute_log_prob, denoiser=standin.flux_denoiser)
        return _keep_rows(out, len(timesteps), keep_positions)


class FluxStandinEngine(_StandinFamilyEngine):
    PLAN = FluxStandinPlan


class QwenStandinPlan:
    def __init__(self, engine, key):
        self.engine = engine
        self.batch, self.n_cfg, self.h, self.w, self.n_text, self.max_steps = key

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance_scale, init_latents, storage_dtype, step_noise, embeds, lens=None,
                keep_positions=None, compute_log_prob=True):
        from oracle import qwen_ref as Q
        from oracle import standin
        B = self.batch
        self.engine.calls.append(("rollout", dict(N=len(timesteps), lens=list(lens), n_cfg=self.n_cfg)))
        lens = [int(n) for n in lens]
        # the plugin hands over [negative | positive] halves padded to the plan's text length; the reference pads each call to its own
        # batch maximum (`_pad_batch_prompt`): cut back so that the stand-in sees tensors of the reference's shapes
        if self.n_cfg == 2:
            nl_, pl_ = lens[:B], lens[B:]
            neg, pos = embeds[:B, :max(nl_)], embeds[B:, :max(pl_)]
        else:
            nl_, pl_, neg, pos = None, lens, None, embeds[:, :max(lens)]
        out = Q.rollout(None, None, pos, pl_, neg, nl_, guidance_scale, init_latents, step_noise, torch.tensor(timesteps, dtype=torch.float32),
                        torch.tensor(sigmas, dtype=torch.float32), list(noise_levels), self.h // 2, self.w // 2, storage_dtype,
                        dynamics_type=dynamics, compute_log_prob=compute_log_prob, denoiser=standin.qwen_denoiser)
        return _keep_rows(out, len(timesteps), keep_positions)


class QwenStandinEngine(_StandinFamilyEngine):
    PLAN = QwenStandinPlan


class WanStandinPlan:
    def __init__(self, engine, key):
        self.engine = engine
        self.batch, self.n_cfg, self.T, self.h, self.w, self.n_text, self.max_steps = key

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance, init_latents, storage_dtype, step_noise, prompt_embeds,
                neg_embeds=None, keep_positions=None, compute_log_prob=True):
        from functools import partial
        from oracle import standin
        from oracle import wan_ref as W
        self.engine.calls.append(("rollout", dict(N=len(timesteps), n_cfg=self.n_cfg)))
        if step_noise is None:
            step_noise = torch.zeros((len(timesteps),) + tuple(init_latents.shape))
        out = W.rollout(None, None, prompt_embeds, neg_embeds, guidance, init_latents, step_noise, torch.tensor(timesteps).long(),
                        torch.tensor(sigmas, dtype=torch.float32), list(noise_levels), storage_dtype, dynamics_type=dynamics,
                        compute_log_prob=compute_log_prob, denoiser=partial(standin.wan_denoiser, expert=0))
        return _keep_rows(out, len(timesteps), keep_positions)


class WanStandinEngine(_StandinFamilyEngine):
    PLAN = WanStandinPlan


# --------------------------------------------------------------------------------------- CPU stand-in for the fused step kernel
def oracle_sde_step(v_text, v_uncond, guidance, latents, sigma, sigma_next, eta, sigma_max, dynamics: str, noise=None, next_latents=None,
                    compute_log_prob=True, want=("next_latents", "next_latents_mean", "noise_pred", "std_dev_t", "dt")):
    """Signature of `mi355_flow.engine.sde_step` (the fused CFG + SDE-step + log-prob kernel, mi355_sde_step) computed by the ORACLE step
    (`oracle.scheduler_ref.sde_step`, bit-exact against the reference scheduler's fixtures; the HIP kernel is pinned against the same
    fixtures on the GPU).  Lets the per-step host paths of the FLUX / Qwen / Wan mixins run on a CPU-only box."""
    from oracle import rollout_ref as R
    from oracle import scheduler_ref as S
    v = v_text if v_uncond is None else R.cfg_combine_bf16(v_uncond, v_text, float(guidance))
    if torch.is_tensor(eta):
        eta = float(eta.reshape(-1)[0])
    out = S.sde_step(v.to(torch.bfloat16) if v.dtype != torch.float32 else v, latents, torch.as_tensor(sigma, dtype=torch.float32),
                     torch.as_tensor(sigma_next, dtype=torch.float32), float(eta), dynamics_type=dynamics, sigma_max=sigma_max,
                     variance_noise=noise, next_latents=next_latents, compute_log_prob=compute_log_prob)
    o = types.SimpleNamespace(**{k: out.get(k) for k in ("next_latents", "next_latents_mean", "noise_pred", "log_prob", "std_dev_t", "dt")})
    o.next_storage = S.cast_latents(out["next_latents"], latents.dtype)
    return o


class WanStandinPlanStepwise(WanStandinPlan):
    """+ `transformer_forward` (what the per-step paths call): [negative | positive] halves through the stand-in of THIS engine's expert."""

    @property
    def engine_expert(self):
        return getattr(self.engine, "expert", 0)

    def transformer_forward(self, latents, t, enc_a, enc_b=None):
        from oracle import standin
        B = latents.shape[0]
        self.engine.calls.append(("transformer_forward", dict(expert=self.engine_expert, n_cfg=1 if enc_b is None else 2, t=float(t.reshape(-1)[0]))))
        tt = t.reshape(-1)[0].long().expand(B)
        halves = [standin.wan_denoiser(latents.to(torch.bfloat16), tt, e, self.engine_expert) for e in ([enc_a] if enc_b is None else [enc_a, enc_b])]
        return torch.cat(halves)


class WanStandinEngineStepwise(_StandinFamilyEngine):
    PLAN = WanStandinPlanStepwise
    _count = 0

    def __init__(self, cfg):
        super().__init__(cfg)
        self.expert = type(self)._count % 2          # the plugin builds the high-noise expert's engine first, then the low-noise one
        type(self)._count += 1


class FluxStandinPlanModel(FluxStandinPlan):
    """FLUX plan double at the MODEL level: `transformer_forward(latents, t_model, guidance_model, pe, pp)` receives the values the product's
    host code computed for the network (`oracle.standin.flux_denoiser_model`); the fused rollout goes through the oracle loop with
    `flux_transformer_call` (the adapter-level arguments -> the model's own first arithmetic -> the same stand-in)."""

    def transformer_forward(self, latents, t_model, guidance_model, prompt_embeds, pooled):
        from oracle import standin
        self.engine.calls.append(("transformer_forward", dict(t=float(t_model.reshape(-1)[0]))))
        return standin.flux_denoiser_model(latents, t_model, guidance_model, pooled, prompt_embeds, self.h // 2, self.w // 2)

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance_scale, init_latents, storage_dtype, step_noise, prompt_embeds, pooled,
                keep_positions=None, compute_log_prob=True):
        from oracle import flux_ref as FR
        from oracle import standin
        self.engine.calls.append(("rollout", dict(N=len(timesteps))))
        if step_noise is None:
            step_noise = torch.zeros((len(timesteps),) + tuple(init_latents.shape))
        out = FR.rollout(None, None, prompt_embeds, pooled, guidance_scale, init_latents, step_noise, torch.tensor(timesteps, dtype=torch.float32),
                         torch.tensor(sigmas, dtype=torch.float32), list(noise_levels),
                         FR.prepare_img_ids(self.h // 2, self.w // 2).to(init_latents.dtype), storage_dtype, dynamics_type=dynamics,
                         compute_log_prob=compute_log_prob, denoiser=standin.flux_transformer_call)
        return _keep_rows(out, len(timesteps), keep_positions)


class FluxStandinEngineModel(_StandinFamilyEngine):
    PLAN = FluxStandinPlanModel


class FluxTrainPlanModel(FluxStandinPlanModel):
    """+ the native training API of `mi355_flow.flux.FluxPlan` (mi355_flux_forward_train / mi355_flux_backward): the forward is the SAME
    stand-in as the no-grad one (bit-identical velocity, like the real engine), plus a dependence on the bound weights so that an optimizer
    step moves the policy; the backward writes d loss / d w = <dv, d v / d w> into the registered fp32 buffers."""

    def _signal(self):
        eng = self.engine
        return sum(float(eng.bound[n].float().mean()) for n in eng.signal_names()) if eng.bound else 0.0

    def transformer_forward(self, latents, t_model, guidance_model, prompt_embeds, pooled):
        v = super().transformer_forward(latents, t_model, guidance_model, prompt_embeds, pooled)
        return (v.float() + self._signal() * latents.float()).to(torch.bfloat16)

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance_scale, init_latents, storage_dtype, step_noise, prompt_embeds, pooled,
                keep_positions=None, compute_log_prob=True):
        from oracle import flux_ref as FR
        from oracle import standin
        self.engine.calls.append(("rollout", dict(N=len(timesteps))))
        if step_noise is None:
            step_noise = torch.zeros((len(timesteps),) + tuple(init_latents.shape))
        sig = self._signal()

        def net(hidden_states=None, **kw):           # the same network as transformer_forward: stand-in + the weight-dependent term
            return (standin.flux_transformer_call(hidden_states=hidden_states, **kw).float() + sig * hidden_states.float()).to(torch.bfloat16)
        out = FR.rollout(None, None, prompt_embeds, pooled, guidance_scale, init_latents, step_noise, torch.tensor(timesteps, dtype=torch.float32),
                         torch.tensor(sigmas, dtype=torch.float32), list(noise_levels),
                         FR.prepare_img_ids(self.h // 2, self.w // 2).to(init_latents.dtype), storage_dtype, dynamics_type=dynamics,
                         compute_log_prob=compute_log_prob, denoiser=net)
        return _keep_rows(out, len(timesteps), keep_positions)

    def forward_train(self, latents, t_model, guidance_model, prompt_embeds, pooled):
        self.engine.calls.append(("forward_train", dict(t=float(t_model.reshape(-1)[0]))))
        self._train_serial = getattr(self, "_train_serial", 0) + 1
        self._stash = latents.float().clone()
        self.engine.calls.pop()                      # (transformer_forward logs itself)
        v = self.transformer_forward(latents, t_model, guidance_model, prompt_embeds, pooled)
        self.engine.calls[-1] = ("forward_train", self.engine.calls[-1][1])
        return v

    def backward(self, dv):
        eng = self.engine
        eng.calls.append(("backward", dict(serial=self._train_serial)))
        up = float((dv.float() * self._stash).sum())            # d v / d signal = latents; signal = sum of the means of the signal tensors
        for name, buf in eng.grad_bufs.items():
            buf += up / buf.numel()


class FluxTrainEngineModel(_StandinFamilyEngine):
    """FLUX.1 engine double WITH the native backward's host API (grad_supported / set_grad / clear_grads), scope = the block linears."""
    PLAN = FluxTrainPlanModel

    def __init__(self, cfg):
        super().__init__(cfg)
        self.grad_bufs = {}

    def signal_names(self):
        return [n for n in self._names if n in self.bound and self.grad_supported(n)]

    def grad_supported(self, name) -> int:
        return 1 if (name in self._names and ".attn.to_" in name) else 0

    def set_grad(self, name, t) -> None:
        assert self.grad_supported(name) and t.dtype in (torch.float32, torch.bfloat16)
        self.grad_bufs[name] = t

    def clear_grads(self) -> None:
        self.grad_bufs = {}


def oracle_sde_step_bwd(v_text, v_uncond, guidance, latents, next_latents, sigma, sigma_next, eta, sigma_max, dynamics: str, compute_log_prob,
                        g_log_prob=None, g_noise_pred=None, g_mean=None):
    """Signature of `mi355_flow.engine.sde_step_bwd` computed by torch autograd through the ORACLE step (`oracle.scheduler_ref.sde_step` is
    plain differentiable torch): d loss / d v for the CPU-only plugin tests -- [uncond | text] when a CFG pair is given (the combine
    `u + g (c - u)` differentiated in fp32)."""
    from oracle import scheduler_ref as S
    if torch.is_tensor(eta):
        eta = float(eta.reshape(-1)[0])
    with torch.enable_grad():                    # (called from inside an autograd Function's backward, where grad mode is off)
        return _oracle_sde_step_bwd(S, v_text, v_uncond, float(guidance), latents, next_latents, sigma, sigma_next, eta, sigma_max, dynamics,
                                    compute_log_prob, g_log_prob, g_noise_pred, g_mean)


def _oracle_sde_step_bwd(S, v_text, v_uncond, guidance, latents, next_latents, sigma, sigma_next, eta, sigma_max, dynamics, compute_log_prob, g_log_prob,
                         g_noise_pred, g_mean):
    vt = v_text.float().detach().requires_grad_(True)
    vu = v_uncond.float().detach().requires_grad_(True) if v_uncond is not None else None
    v = vt if vu is None else vu + guidance * (vt - vu)
    out = S.sde_step(v, latents, torch.as_tensor(sigma, dtype=torch.float32), torch.as_tensor(sigma_next, dtype=torch.float32), float(eta),
                     dynamics_type=dynamics, sigma_max=sigma_max, next_latents=next_latents, compute_log_prob=bool(compute_log_prob))
    loss = v.sum() * 0.0
    if g_log_prob is not None and out["log_prob"] is not None:
        loss = loss + (g_log_prob.float() * out["log_prob"]).sum()
    if g_noise_pred is not None:
        loss = loss + (g_noise_pred.float() * out["noise_pred"]).sum()
    if g_mean is not None:
        loss = loss + (g_mean.float() * out["next_latents_mean"]).sum()
    if vu is None:
        (dv,) = torch.autograd.grad(loss, vt)
        return dv
    du, dt_ = torch.autograd.grad(loss, (vu, vt))
    return torch.cat([du, dt_])


class WanTrainPlanModel(WanStandinPlanStepwise):
    """+ the native training API of `mi355_flow.wan.WanPlan` (mi355_wan_forward_train / mi355_wan_backward): the forward is the SAME stand-in as
    the no-grad one plus a dependence on the bound weights; the backward writes d loss / d w = <dv, d v / d w> into the registered buffers
    (as `FluxTrainPlanModel`).  v = [negative | positive] halves when n_cfg == 2; dv likewise."""

    def _signal(self):
        eng = self.engine
        return sum(float(eng.bound[n].float().mean()) for n in eng.signal_names()) if eng.bound else 0.0

    def transformer_forward(self, latents, t, enc_a, enc_b=None):
        v = super().transformer_forward(latents, t, enc_a, enc_b)
        rep = 1 if enc_b is None else 2
        return (v.float() + self._signal() * latents.to(torch.bfloat16).float().repeat(rep, 1, 1, 1, 1)).to(torch.bfloat16)   # (the network sees bf16 latents)

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance, init_latents, storage_dtype, step_noise, prompt_embeds,
                neg_embeds=None, keep_positions=None, compute_log_prob=True):
        from oracle import standin
        from oracle import wan_ref as W
        self.engine.calls.append(("rollout", dict(N=len(timesteps), n_cfg=self.n_cfg)))
        if step_noise is None:
            step_noise = torch.zeros((len(timesteps),) + tuple(init_latents.shape))
        sig = self._signal()

        def net(x, t, enc):                          # the same network as transformer_forward: stand-in + the weight-dependent term
            return (standin.wan_denoiser(x, t, enc, 0).float() + sig * x.float()).to(torch.bfloat16)
        out = W.rollout(None, None, prompt_embeds, neg_embeds, guidance, init_latents, step_noise, torch.tensor(timesteps).long(),
                        torch.tensor(sigmas, dtype=torch.float32), list(noise_levels), storage_dtype, dynamics_type=dynamics,
                        compute_log_prob=compute_log_prob, denoiser=net)
        return _keep_rows(out, len(timesteps), keep_positions)

    def forward_train(self, latents, t, enc_a, enc_b=None):
        self._train_serial = getattr(self, "_train_serial", 0) + 1
        rep = 1 if enc_b is None else 2
        self._stash = latents.to(torch.bfloat16).float().repeat(rep, 1, 1, 1, 1).clone()
        v = self.transformer_forward(latents, t, enc_a, enc_b)
        self.engine.calls[-1] = ("forward_train", self.engine.calls[-1][1])
        return v

    def backward(self, dv):
        eng = self.engine
        eng.calls.append(("backward", dict(serial=self._train_serial)))
        assert dv.shape == self._stash.shape, (dv.shape, self._stash.shape)
        up = float((dv.float() * self._stash).sum())
        for name, buf in eng.grad_bufs.items():
            buf += up / buf.numel()


class WanTrainEngineModel(FluxTrainEngineModel):
    """Wan engine double WITH the native backward's host API; scope = the attention projections (as the FLUX.1 double)."""
    PLAN = WanTrainPlanModel
    expert = 0

    def grad_supported(self, name) -> int:
        return 1 if (name in self._names and (".attn1.to_" in name or ".attn2.to_" in name)) else 0


class QwenStandinPlanModel(QwenStandinPlan):
    """Qwen-Image plan double at the MODEL level: `transformer_forward(latents, t_model, embeds, lens, guidance)` = the prediction the
    scheduler sees (norm-rescaled true CFG over the [negative | positive] halves when n_cfg == 2)."""

    def transformer_forward(self, latents, t_model, embeds, lens=None, guidance_scale=1.0, return_raw=False):
        from oracle import qwen_ref as Q
        from oracle import standin
        B = self.batch
        lens = [int(n) for n in lens]
        self.engine.calls.append(("transformer_forward", dict(t=float(t_model.reshape(-1)[0]), n_cfg=self.n_cfg)))
        if self.n_cfg == 2:
            neg = standin.qwen_denoiser_model(latents, t_model, embeds[:B], lens[:B])
            pos = standin.qwen_denoiser_model(latents, t_model, embeds[B:], lens[B:])
            return Q.cfg_rescale_bf16(neg, pos, float(guidance_scale))
        return standin.qwen_denoiser_model(latents, t_model, embeds, lens)

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance_scale, init_latents, storage_dtype, step_noise, embeds, lens=None,
                keep_positions=None, compute_log_prob=True):
        from oracle import qwen_ref as Q
        from oracle import standin
        B = self.batch
        self.engine.calls.append(("rollout", dict(N=len(timesteps), lens=list(lens), n_cfg=self.n_cfg)))
        lens = [int(n) for n in lens]
        if self.n_cfg == 2:
            nl_, pl_ = lens[:B], lens[B:]
            neg, pos = embeds[:B, :max(nl_)], embeds[B:, :max(pl_)]
        else:
            nl_, pl_, neg, pos = None, lens, None, embeds[:, :max(lens)]
        out = Q.rollout(None, None, pos, pl_, neg, nl_, guidance_scale, init_latents, step_noise, torch.tensor(timesteps, dtype=torch.float32),
                        torch.tensor(sigmas, dtype=torch.float32), list(noise_levels), self.h // 2, self.w // 2, storage_dtype,
                        dynamics_type=dynamics, compute_log_prob=compute_log_prob, denoiser=standin.qwen_transformer_call)
        return _keep_rows(out, len(timesteps), keep_positions)


class QwenStandinEngineModel(_StandinFamilyEngine):
    PLAN = QwenStandinPlanModel


class QwenTrainPlanModel(QwenStandinPlanModel):
    """+ the native training API of `mi355_flow.qwen.QwenPlan` (mi355_qwen_forward_train / mi355_qwen_backward), single-branch (n_cfg == 1)
    plans: the forward is the SAME stand-in as the no-grad one plus a dependence on the bound weights, so that an optimizer step moves the
    policy; the backward writes d loss / d w = <dv, d v / d w> into the registered fp32 buffers (as `FluxTrainPlanModel`)."""

    def _signal(self):
        eng = self.engine
        return sum(float(eng.bound[n].float().mean()) for n in eng.signal_names()) if eng.bound else 0.0

    def transformer_forward(self, latents, t_model, embeds, lens=None, guidance_scale=1.0, return_raw=False):
        assert self.n_cfg == 1
        v = super().transformer_forward(latents, t_model, embeds, lens, guidance_scale)
        return (v.float() + self._signal() * latents.float()).to(torch.bfloat16)

    def rollout(self, timesteps, sigmas, noise_levels, dynamics, guidance_scale, init_latents, storage_dtype, step_noise, embeds, lens=None,
                keep_positions=None, compute_log_prob=True):
        from oracle import qwen_ref as Q
        from oracle import standin
        assert self.n_cfg == 1
        self.engine.calls.append(("rollout", dict(N=len(timesteps), lens=list(lens), n_cfg=self.n_cfg)))
        lens = [int(n) for n in lens]
        sig = self._signal()

        def net(hidden_states=None, **kw):           # the same network as transformer_forward: stand-in + the weight-dependent term
            return (standin.qwen_transformer_call(hidden_states=hidden_states, **kw).float() + sig * hidden_states.float()).to(torch.bfloat16)
        out = Q.rollout(None, None, embeds[:, :max(lens)], lens, None, None, guidance_scale, init_latents, step_noise,
                        torch.tensor(timesteps, dtype=torch.float32), torch.tensor(sigmas, dtype=torch.float32), list(noise_levels), self.h // 2,
                        self.w // 2, storage_dtype, dynamics_type=dynamics, compute_log_prob=compute_log_prob, denoiser=net)
        return _keep_rows(out, len(timesteps), keep_positions)

    def forward_train(self, latents, t_model, embeds, lens=None, guidance_scale=1.0):
        self._train_serial = getattr(self, "_train_serial", 0) + 1
        self._stash = latents.float().clone()
        v = self.transformer_forward(latents, t_model, embeds, lens, guidance_scale)
        self.engine.calls[-1] = ("forward_train", self.engine.calls[-1][1])
        return v

    def backward(self, dv):
        eng = self.engine
        eng.calls.append(("backward", dict(serial=self._train_serial)))
        up = float((dv.float() * self._stash).sum())
        for name, buf in eng.grad_bufs.items():
            buf += up / buf.numel()


class QwenTrainEngineModel(FluxTrainEngineModel):
    """Qwen-Image engine double WITH the native backward's host API; scope = the attention projections (as the FLUX.1 double)."""
    PLAN = QwenTrainPlanModel


# --------------------------------------------------------------------------------------- a miniature `peft` for the reference's apply_lora()
class MiniLoraConfig:
    def __init__(self, r=8, lora_alpha=16, init_lora_weights=True, target_modules=(), **unused):
        self.r, self.lora_alpha, self.target_modules = int(r), float(lora_alpha), list(target_modules)


class MiniPeftModel(FakePeftModel):
    """What the reference's `apply_lora` / `use_ref_parameters` ask of `peft.PeftModel` (models/abc.py:856-950, :556-583): adapter registry,
    `set_adapter`, `disable_adapter()`; every non-LoRA parameter frozen, B = 0 at initialisation (the delta starts at zero, as peft's)."""

    def __init__(self, inner, cfg: "MiniLoraConfig"):
        wrapped = []
        for path, mod in list(inner.named_modules()):
            if any(path.endswith(t) for t in cfg.target_modules) and hasattr(mod, "weight") and not hasattr(mod, "base_layer"):
                parent = inner.get_submodule(path.rsplit(".", 1)[0])
                lay = FakeLoraLinear(mod, r=cfg.r, alpha=cfg.lora_alpha, seed=len(wrapped))
                with torch.no_grad():
                    lay.lora_B["default"].weight.zero_()
                setattr(parent, path.rsplit(".", 1)[1], lay)
                wrapped.append(path)
        assert wrapped, f"no module matches {cfg.target_modules}"
        super().__init__(inner)
        self.wrapped = wrapped
        for n, p in self.named_parameters():
            p.requires_grad_("lora_" in n)
        self.active_adapter = "default"

    def set_adapter(self, name):
        self.active_adapter = name

    def add_adapter(self, name, cfg):
        raise NotImplementedError

    def delete_adapter(self, name):
        raise NotImplementedError


def mini_get_peft_model(model, cfg):
    return MiniPeftModel(model, cfg)
