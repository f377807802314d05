"""Pins the rollout control-flow restatement `oracle/rollout_ref.py` against the REFERENCE's own adapter code.

`tests/golden/rollout_control_flow.npz` holds what the reference's `SD3_5Adapter.inference()` returned on CPU
(src/flow_factory/models/stable_diffusion/sd3_5.py:176-448, imported whole by `oracle/ref_package.py`; generator
`oracle/make_rollout_golden.py`) with the closed-form `oracle.standin.denoiser` in place of the transformer.  The oracle's rollout runs
on the same stand-in, the same seed and the same prompt tensors: trajectory positions, log-probs and the per-step means must agree
BIT FOR BIT -- RNG draw order and dtypes, the timestep rounding, CFG batch order and bf16 arithmetic, the scheduler's inputs, the
storage-dtype round trips, which positions / SDE steps are kept.  (The network body itself stays an unpinned restatement of
un-vendored diffusers; everything around it is pinned here.)  In the build container the fixture is also re-generated live from
/root/reference and compared with the committed file."""
import os

import numpy as np
import pytest
import torch

from oracle import rollout_ref as R
from oracle import standin

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "rollout_control_flow.npz")
CASES = {"flow_sde_cfg_fp16": ("Flow-SDE", 4.5), "cps_nocfg_bf16": ("CPS", 1.0), "dance_cfg_native": ("Dance-SDE", 3.0),
         "eval_ode_cfg_fp16": ("Flow-SDE", 4.5)}
DT = {0: torch.float16, 1: torch.bfloat16, 2: torch.float32}


def _case(blob, name):
    return {k.split("/", 1)[1]: torch.from_numpy(np.asarray(blob[k])) for k in blob.files if k.startswith(name + "/")}


def _check(name, ref):
    dyn, gs = CASES[name]
    storage = DT[int(ref["latents_dtype"])]
    B, _, C, h, w = ref["all_latents"].shape
    N = ref["timesteps"].numel()
    pe, pp, ne, npl = (ref[k].bfloat16() for k in ("pe", "pp", "ne", "npl"))
    torch.manual_seed(int(ref["seed"]))                      # the reference's draws: prepare_latents (transformer dtype), then one fp32 per step
    init, noise = R.draw_rollout_noise(B, C, h, w, N, torch.bfloat16, None)
    nl = [float(x) for x in ref["noise_levels"]]
    out = R.rollout(None, None, pe, pp, ne if gs > 1 else None, npl if gs > 1 else None, gs, init, noise, ref["timesteps"], ref["sigmas"], nl,
                    storage, dynamics_type=dyn, compute_log_prob="log_probs" in ref, denoiser=standin.denoiser, is_eval=bool(ref["is_eval"]))
    lmap = ref["latent_index_map"].long()
    kept = [p for p in range(N + 1) if lmap[p] >= 0]
    assert kept, name
    for p in kept:
        got, want = out["all_latents"][p].float(), ref["all_latents"][:, lmap[p]]
        assert torch.equal(got, want), (name, "latents at position", p, float((got - want).abs().max()))
    if "log_probs" in ref:
        pmap = ref["log_prob_index_map"].long()
        sde = [i for i in range(N) if pmap[i] >= 0]
        assert sde == [i for i in range(N) if nl[i] > 0] and not bool(ref["is_eval"]), (name, sde, nl)
        for i in sde:
            assert torch.equal(out["log_probs"][i], ref["log_probs"][:, pmap[i]]), (name, "log-prob of step", i)
    cmap = ref["callback_index_map"].long()
    for i in range(N):
        if cmap[i] >= 0:
            assert torch.equal(out["next_latents_means"][i].float(), ref["next_latents_mean"][:, cmap[i]]), (name, "mean of step", i)
    if "replay_log_prob" in ref:
        # optimize()'s replay of the stored transitions (trainers/grpo.py:229-263): the reference's own replay reproduces its rollout
        # log-probs exactly (train_inference_consistency.md), and the oracle's replay step reproduces every replay output
        pmap = ref["log_prob_index_map"].long()
        for j, i in enumerate(i for i in range(N) if pmap[i] >= 0):
            assert torch.equal(ref["replay_log_prob"][j], ref["log_probs"][:, pmap[i]]), (name, "the reference's own replay ratio != 1", i)
            x_i, x_n = ref["all_latents"][:, lmap[i]].to(storage), ref["all_latents"][:, lmap[i + 1]].to(storage)
            t_next = ref["timesteps"][i + 1] if i + 1 < N else torch.tensor(0.0)
            o = R.forward_step(None, None, ref["timesteps"][i].expand(B), t_next.expand(B), x_i, pe, pp, ne if gs > 1 else None,
                               npl if gs > 1 else None, gs, noise_level=nl[i], dynamics_type=dyn, sigma_max=float(ref["sigmas"][1]),
                               next_latents=x_n, compute_log_prob=True, denoiser=standin.denoiser)
            for k in ("log_prob", "noise_pred", "next_latents_mean", "std_dev_t", "dt"):
                want = ref["replay_" + k][j]
                assert torch.equal(o[k].float().reshape(want.shape), want), (name, "replay", k, i)
    return len(kept)


def _compare(name, ref, out, N, nl):
    lmap = ref["latent_index_map"].long()
    kept = [p for p in range(N + 1) if lmap[p] >= 0]
    for p in kept:
        got, want = out["all_latents"][p].float(), ref["all_latents"][:, lmap[p]]
        assert torch.equal(got, want), (name, "latents at position", p, float((got - want).abs().max()))
    pmap = ref["log_prob_index_map"].long()
    sde = [i for i in range(N) if pmap[i] >= 0]
    assert sde and sde == [i for i in range(N) if nl[i] > 0], (name, sde, nl)
    for i in sde:
        assert torch.equal(out["log_probs"][i], ref["log_probs"][:, pmap[i]]), (name, "log-prob of step", i)
    cmap = ref["callback_index_map"].long()
    for i in range(N):
        if cmap[i] >= 0:
            assert torch.equal(out["next_latents_means"][i].float(), ref["next_latents_mean"][:, cmap[i]]), (name, "mean of step", i)
    return len(kept)


FLUX_CASES = {"flux_flow_sde_fp16": "Flow-SDE", "flux_dance_native": "Dance-SDE"}


def _check_flux(name, ref):
    """`Flux1Adapter.inference` / `.forward` (reference models/flux/flux1.py:151-346) vs `oracle.flux_ref.rollout`: the latents are drawn
    in the PROMPT EMBEDDINGS' dtype as (B, 16, h, w) and packed, the per-step noise in fp32 with the PACKED shape, the transformer sees
    `t / 1000` in fp32 and the guidance scale in the latents' dtype."""
    from oracle import flux_ref as FR
    storage = DT[int(ref["latents_dtype"])]
    B, _, Ni, ch = ref["all_latents"].shape
    N = ref["timesteps"].numel()
    hp = int(ref["img_ids"][:, 1].max()) + 1
    wp = int(ref["img_ids"][:, 2].max()) + 1
    pe, pp = ref["pe"].bfloat16(), ref["pp"].bfloat16()
    torch.manual_seed(int(ref["seed"]))
    init = FR.pack_latents(torch.randn(B, ch // 4, 2 * hp, 2 * wp, dtype=pe.dtype))
    noise = torch.stack([torch.randn(B, Ni, ch, dtype=torch.float32) for _ in range(N)])
    nl = [float(x) for x in ref["noise_levels"]]
    out = FR.rollout(None, None, pe, pp, float(ref["guidance"]), init, noise, ref["timesteps"], ref["sigmas"], nl, ref["img_ids"].to(pe.dtype),
                     storage, dynamics_type=FLUX_CASES[name], denoiser=standin.flux_denoiser)
    return _compare(name, ref, out, N, nl)


@pytest.mark.parametrize("name", sorted(FLUX_CASES))
def test_flux_rollout_oracle_reproduces_the_reference_adapter_bit_for_bit(name):
    assert _check_flux(name, _case(np.load(GOLDEN), name)) >= 2


QWEN_CASES = {"qwen_flow_sde_cfg_ragged": "Flow-SDE", "qwen_cps_nocfg_fp16": "CPS"}


def _check_qwen(name, ref):
    """`QwenImageAdapter.inference` / `.forward` (reference models/qwen_image/qwen_image.py:288-600) vs `oracle.qwen_ref.rollout`: ragged
    prompt lists padded by the reference's own `_pad_batch_prompt`, `timestep / 1000` in the latents' dtype, a cond and an uncond call,
    true CFG rescaled to the norm of the cond prediction in bf16."""
    from oracle import flux_ref as FR
    from oracle import qwen_ref as Q
    storage = DT[int(ref["latents_dtype"])]
    B, _, Ni, ch = ref["all_latents"].shape
    N = ref["timesteps"].numel()
    hp, wp = (int(v) for v in ref["hw"])
    gs = float(ref["guidance"])
    pe, ne = ref["pe"].bfloat16(), ref["ne"].bfloat16()
    torch.manual_seed(int(ref["seed"]))
    init = FR.pack_latents(torch.randn(B, 1, ch // 4, 2 * hp, 2 * wp, dtype=torch.bfloat16)[:, 0])      # transformer dtype
    noise = torch.stack([torch.randn(B, Ni, ch, dtype=torch.float32) for _ in range(N)])
    nl = [float(x) for x in ref["noise_levels"]]
    out = Q.rollout(None, None, pe, ref["lens"].tolist(), ne if gs > 1 else None, ref["nlens"].tolist() if gs > 1 else None, gs, init, noise,
                    ref["timesteps"], ref["sigmas"], nl, hp, wp, storage, dynamics_type=QWEN_CASES[name], denoiser=standin.qwen_denoiser)
    return _compare(name, ref, out, N, nl)


@pytest.mark.parametrize("name", sorted(QWEN_CASES))
def test_qwen_rollout_oracle_reproduces_the_reference_adapter_bit_for_bit(name):
    assert _check_qwen(name, _case(np.load(GOLDEN), name)) >= 2


WAN_CASES = {"wan21_flow_sde_cfg_fp16": "Flow-SDE", "wan22_two_expert_cps": "CPS"}


def _check_wan(name, ref):
    """`Wan2_T2V_Adapter.inference` / `.forward` (reference models/wan/wan2_t2v.py:234-543) on the reference's own `UniPCMultistepSDEScheduler`
    rollout branch vs `oracle.wan_ref.rollout` / `rollout_two_expert`: latents and step noise drawn in fp32, the transformer sees them in its
    own dtype with the INTEGER timestep, cond / uncond passes combined in bf16; Wan2.2: the expert and its guidance scale chosen per step by
    `t >= boundary_ratio * 1000`."""
    from functools import partial
    from oracle import wan_ref as W
    storage = DT[int(ref["latents_dtype"])]
    B, _, C, T, h, w = ref["all_latents"].shape
    N = ref["timesteps"].numel()
    pe, ne = ref["pe"].bfloat16(), ref["ne"].bfloat16()
    torch.manual_seed(int(ref["seed"]))
    init = torch.randn(B, C, T, h, w, dtype=torch.float32)
    noise = torch.stack([torch.randn(B, C, T, h, w, dtype=torch.float32) for _ in range(N)])
    nl = [float(x) for x in ref["noise_levels"]]
    ts = ref["timesteps"].long()
    f0, f1 = partial(standin.wan_denoiser, expert=0), partial(standin.wan_denoiser, expert=1)
    if float(ref["boundary_timestep"]) < 0:
        out = W.rollout(None, None, pe, ne, float(ref["guidance"]), init, noise, ts, ref["sigmas"], nl, storage, dynamics_type=WAN_CASES[name],
                        denoiser=f0)
    else:
        out = W.rollout_two_expert(None, None, None, float(ref["boundary_timestep"]), pe, ne, float(ref["guidance"]), float(ref["guidance_2"]),
                                   init, noise, ts, ref["sigmas"], nl, storage, dynamics_type=WAN_CASES[name], denoisers=(f0, f1))
        assert any(float(t) >= float(ref["boundary_timestep"]) for t in ts) and any(float(t) < float(ref["boundary_timestep"]) for t in ts)
    return _compare(name, ref, out, N, nl)


@pytest.mark.parametrize("name", sorted(WAN_CASES))
def test_wan_rollout_oracle_reproduces_the_reference_adapter_bit_for_bit(name):
    assert _check_wan(name, _case(np.load(GOLDEN), name)) >= 2


@pytest.mark.parametrize("name", sorted(CASES))
def test_rollout_oracle_reproduces_the_reference_adapter_bit_for_bit(name):
    blob = np.load(GOLDEN)
    assert _check(name, _case(blob, name)) >= 2


def test_fixture_is_what_the_reference_produces_now():
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    from oracle import make_rollout_golden as G
    blob = np.load(GOLDEN)
    for cases, run, check in ((CASES, G.run_reference, _check), (FLUX_CASES, G.run_reference_flux, _check_flux),
                              (QWEN_CASES, G.run_reference_qwen, _check_qwen), (WAN_CASES, G.run_reference_wan, _check_wan)):
        for name in sorted(cases):
            live = run(name)
            stored = _case(blob, name)
            assert sorted(live) == sorted(stored), name
            for k, v in live.items():
                assert torch.equal(v.detach().cpu().float(), stored[k].float()), (name, k)
            check(name, {k: v.detach().cpu() for k, v in live.items()})


@pytest.mark.parametrize("name", sorted(CASES))
def test_plugin_host_path_reproduces_the_reference_adapter(name):
    """The PRODUCT's host path against the reference adapter: the Flow-Factory plugin class (`mi355_flow.flow_factory_plugin.
    SD3_5NativeAdapter` = reference adapter + rollout mixin) is built and called exactly like the reference adapter was for the fixture,
    with an engine double that computes (oracle SDE step around the same stand-in network) where libmi355flow.so would.  Everything the
    mixin does on the host -- the RNG draws and their dtypes, the schedule, the per-step noise levels, which positions / SDE steps are
    kept, the per-step callback capture, the reference sample class and its index maps -- must reproduce the reference's samples bit for
    bit; the torch transformer is never called."""
    import sys
    import types
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    from oracle import make_rollout_golden as G
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    saved = (P.Engine, P.VAEDecoder, P.VAEConfig)
    P.Engine, P.VAEDecoder, P.VAEConfig = F.StandinEngine, F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c)
    try:
        live = G.run_reference(name, adapter_base=P.SD3_5NativeAdapter)
    finally:
        P.Engine, P.VAEDecoder, P.VAEConfig = saved
    stored = _case(np.load(GOLDEN), name)
    assert sorted(live) == sorted(stored), name
    for k, v in live.items():
        assert torch.equal(v.detach().cpu().float(), stored[k].float()), (name, k)


@pytest.mark.parametrize("family,name", [("flux", "flux_flow_sde_fp16"), ("flux", "flux_dance_native"), ("qwen", "qwen_flow_sde_cfg_ragged"),
                                         ("qwen", "qwen_cps_nocfg_fp16"), ("wan", "wan21_flow_sde_cfg_fp16"), ("wan", "wan22_ti2v_expand_timesteps")])
@pytest.mark.parametrize("explicit_generator", [False, True], ids=["global-rng", "explicit-generator"])
def test_family_plugin_host_paths_reproduce_the_reference_adapters(family, name, explicit_generator):
    """Same as above for `Flux1NativeAdapter`, `QwenImageNativeAdapter` and `Wan2T2VNativeAdapter` (single transformer): the reference
    adapter and the plugin class are built and called identically (fused rollouts: no per-step callback tensors), the plugin on an engine
    double whose `rollout` is the family's oracle loop around the stand-in network.  Samples must agree bit for bit -- packed / 5-D latent
    draws in the right dtype and order, dynamic shift / UniPC schedules, text padding and lengths, kept positions, sample fields."""
    import sys
    import types
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    import mi355_flow.vae as MV
    from oracle import make_rollout_golden as G
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    run, attr, eng, names, plug = {
        "flux": (G.run_reference_flux, "FluxEngine", F.FluxStandinEngine,
                 ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias", "x_embedder.weight"], "Flux1NativeAdapter"),
        "qwen": (G.run_reference_qwen, "QwenEngine", F.QwenStandinEngine,
                 ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias"], "QwenImageNativeAdapter"),
        "wan": (G.run_reference_wan, "WanEngine", F.WanStandinEngine, ["blocks.0.attn1.to_q.weight", "blocks.0.attn1.to_q.bias"],
                "Wan2T2VNativeAdapter"),
    }[family]
    # (an explicit `generator` only feeds `prepare_latents` in the reference; the per-step noise stays on the global generator)
    want = run(name, callbacks=False, explicit_generator=explicit_generator)      # the reference's own adapter, live
    saved = (getattr(P, attr), P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder)
    eng.NAMES = names
    setattr(P, attr, eng)
    P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c), F.FakeVideoVAEDecoder
    try:
        got = run(name, adapter_base=getattr(P, plug), callbacks=False, explicit_generator=explicit_generator)
    finally:
        setattr(P, attr, saved[0])
        P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = saved[1:]
    assert sorted(got) == sorted(want), name
    for k, v in want.items():
        assert torch.equal(got[k].detach().cpu().float(), v.detach().cpu().float()), (name, k)
    assert want["all_latents"].shape[1] >= 2 and want["log_probs"].numel() > 0


def _sample_fields(samples):
    """Every tensor a sample carries (incl. extra_kwargs), stacked over the batch, plus the scalar / list fields."""
    import dataclasses
    out = {}
    for f in dataclasses.fields(samples[0]):
        vals = [getattr(s, f.name) for s in samples]
        if f.name == "extra_kwargs":
            for k in sorted(vals[0]):
                out["extra." + k] = [v[k] for v in vals]
        else:
            out[f.name] = vals
    return out


def _assert_same_samples(got, want, ctx):
    a, b = _sample_fields(got), _sample_fields(want)
    assert sorted(a) == sorted(b), (ctx, sorted(set(a) ^ set(b)))
    for k in a:
        for x, y in zip(a[k], b[k]):
            if torch.is_tensor(y):
                assert torch.is_tensor(x) and x.dtype == y.dtype and x.shape == y.shape, (ctx, k, getattr(x, "dtype", None), y.dtype)
                assert torch.equal(torch.nan_to_num(x.float()), torch.nan_to_num(y.float())), (ctx, k)
            else:
                assert x == y, (ctx, k, x, y)


@pytest.mark.parametrize("trial", range(40))
def test_differential_sweep_plugin_vs_reference_adapter(trial):
    """Differential sweep over the options of `inference()`: dynamics, CFG on / off / requested-without-negatives, storage dtype, step
    count, SDE-step windows, trajectory selections ('all', None, explicit lists with negative indices), log-probs on / off, callback keys,
    global vs explicit CPU generator, eval mode.  For every drawn combination the reference's `SD3_5Adapter` and the plugin class (engine
    double computing the oracle step around the stand-in network) are built and called identically; EVERY field of every returned sample
    -- tensors with dtype and shape, index maps, extra_kwargs, prompts, sizes -- must be equal."""
    import random
    import sys
    import types
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    from oracle import make_rollout_golden as G
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    rnd = random.Random(7700 + trial)
    dyn = rnd.choice(["Flow-SDE", "Dance-SDE", "CPS", "ODE"])
    storage = rnd.choice(["fp16", "bf16", None])
    N = rnd.choice([3, 4, 6, 7])
    window = sorted(rnd.sample(range(N - 1), rnd.randint(1, N - 1)))
    n_sde = rnd.randint(1, len(window))
    is_eval = rnd.random() < 0.2
    cfg_mode = rnd.choice(["cfg", "nocfg", "cfg_without_negatives"])
    gs = 1.0 if cfg_mode == "nocfg" else rnd.choice([2.0, 4.5])
    # (eval mode + log-probs: the reference collects what its expression gives at eta = 0 -- NaN under Flow- / Dance-SDE; the CPS residual
    #  is not produced by the engine and raises)
    clp = dyn != "ODE" and not (is_eval and dyn == "CPS") and rnd.random() < 0.8
    # (log-probs with a selection that keeps no SDE step make the reference's own collector stack an empty list: not drawn)
    traj = rnd.choice(["all", "train"]) if clp else rnd.choice(["all", None, [0, -1], [-1], [1, 2, -2]])
    callbacks = rnd.choice([[], ["noise_level"], ["next_latents_mean"], ["noise_pred", "std_dev_t", "dt", "noise_level"]])
    explicit_gen = rnd.choice([None, None, "one", "per-sample list"])      # evaluate() passes one CPU generator per prompt
    ctx = dict(trial=trial, dyn=dyn, storage=storage, N=N, window=window, n_sde=n_sde, is_eval=is_eval, cfg=cfg_mode, gs=gs, clp=clp,
               traj=traj, callbacks=callbacks, explicit_gen=explicit_gen)
    g = torch.Generator().manual_seed(100 + trial)
    Bq = rnd.choice([1, 2, 3])
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16()     # noqa: E731
    pe, pp, ne, npl = mk(Bq, 7, 128), mk(Bq, 128), mk(Bq, 7, 128), mk(Bq, 128)

    def run(base):
        from flow_factory.utils.trajectory_collector import compute_trajectory_indices
        ad = G.build_sd3(base, dyn, storage, window, n_sde, 0.7, is_eval, seed=trial)
        ti = compute_trajectory_indices(train_timestep_indices=ad.scheduler.train_timesteps, num_inference_steps=N) if traj == "train" else traj
        torch.manual_seed(4242 + trial)
        kw = dict(prompt=[f"p{i}" for i in range(Bq)], prompt_ids=torch.arange(Bq * 3).reshape(Bq, 3), height=64, width=96,
                  num_inference_steps=N, guidance_scale=gs, prompt_embeds=pe, pooled_prompt_embeds=pp, compute_log_prob=clp,
                  trajectory_indices=ti, extra_call_back_kwargs=list(callbacks),
                  generator=(None if explicit_gen is None else torch.Generator().manual_seed(9 + trial) if explicit_gen == "one" else
                             [torch.Generator().manual_seed(9 + trial + 17 * i) for i in range(Bq)]))
        if cfg_mode == "cfg":
            kw.update(negative_prompt_embeds=ne, negative_pooled_prompt_embeds=npl, negative_prompt_ids=torch.zeros(Bq, 3, dtype=torch.long))
        return ad.inference(**kw)

    try:
        want, ref_error = run(None), None
    except Exception as e:          # noqa: BLE001 -- an option combination the reference itself cannot serve
        want, ref_error = None, e
    saved = (P.Engine, P.VAEDecoder, P.VAEConfig)
    P.Engine, P.VAEDecoder, P.VAEConfig = F.StandinEngine, F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c)
    try:
        if ref_error is not None:
            # e.g. log-probs requested while the trajectory selection keeps no SDE step: the reference's collector stacks an empty
            # list (RuntimeError); the plugin returns no log-probs instead.  Nothing to compare.
            pytest.skip(f"the reference itself cannot serve this combination: {ref_error!r}")
        got = run(P.SD3_5NativeAdapter)
    finally:
        P.Engine, P.VAEDecoder, P.VAEConfig = saved
    # (the decoded image comes from a VAE double on both sides: compare everything else)
    for s_ in list(got) + list(want):
        s_.image = None
    _assert_same_samples(got, want, ctx)


@pytest.mark.parametrize("trial", range(40))
def test_differential_sweep_forward_plugin_vs_reference_adapter(trial):
    """The single-step API, `forward()` (no-grad; the rollout step, the KL / old-policy forwards of the trainers, the replay's value path):
    scalar or per-sample `t`, `t_next` given or derived from the schedule, `noise_level` given or inferred from sigma, a sampling step
    (fresh noise from the global generator) or the replay of a stored transition, any subset of `return_kwargs`, CFG on / off / without
    negatives, eval mode -- every returned field must equal the reference adapter's, with dtype and shape."""
    import random
    import sys
    import types
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    from oracle import make_rollout_golden as G
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    rnd = random.Random(9100 + trial)
    dyn = rnd.choice(["Flow-SDE", "Dance-SDE", "CPS", "ODE"])
    storage = rnd.choice(["fp16", "bf16", None])
    N = rnd.choice([4, 6, 8])
    window = sorted(rnd.sample(range(N - 1), rnd.randint(1, N - 1)))
    is_eval = rnd.random() < 0.15
    cfg_mode = rnd.choice(["cfg", "nocfg", "cfg_without_negatives"])
    gs = 1.0 if cfg_mode == "nocfg" else rnd.choice([2.0, 4.5])
    Bq = rnd.choice([1, 2, 3])
    step = rnd.randrange(N - 1) if dyn == "CPS" else rnd.randrange(N)          # (CPS at the last step: sigma_next = 0, a delta)
    per_sample_t = rnd.random() < 0.5
    give_t_next = rnd.random() < 0.6
    noise_level = rnd.choice([None, 0.0, 0.7])
    replay = rnd.random() < 0.5
    clp = dyn != "ODE" and not is_eval and rnd.random() < 0.7 and (noise_level is None or noise_level > 0)
    keys = ["noise_pred", "next_latents", "next_latents_mean", "std_dev_t", "dt", "log_prob"]
    return_kwargs = rnd.sample(keys, rnd.randint(1, len(keys)))
    ctx = dict(trial=trial, dyn=dyn, storage=storage, N=N, window=window, is_eval=is_eval, cfg=cfg_mode, gs=gs, B=Bq, step=step,
               per_sample_t=per_sample_t, give_t_next=give_t_next, noise_level=noise_level, replay=replay, clp=clp, return_kwargs=return_kwargs)
    g = torch.Generator().manual_seed(300 + trial)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16()     # noqa: E731
    pe, pp, ne, npl = mk(Bq, 7, 128), mk(Bq, 128), mk(Bq, 7, 128), mk(Bq, 128)
    sdt = {"fp16": torch.float16, "bf16": torch.bfloat16, None: torch.bfloat16}[storage]
    x = torch.randn(Bq, 16, 8, 12, generator=g).to(sdt)
    x_next = (x.float() * 0.9 + 0.1 * torch.randn(Bq, 16, 8, 12, generator=g)).to(sdt)

    def run(base):
        ad = G.build_sd3(base, dyn, storage, window, 1, 0.7, is_eval, seed=trial)
        ad.scheduler.set_timesteps(N) if not hasattr(ad.scheduler, "_mi355") else None
        from flow_factory.scheduler import set_scheduler_timesteps
        ts = set_scheduler_timesteps(scheduler=ad.scheduler, num_inference_steps=N, seq_len=24, device=torch.device("cpu"))
        t = ts[step].expand(Bq).clone() if per_sample_t else ts[step]
        t_next = (ts[step + 1] if step + 1 < N else torch.tensor(0.0))
        if per_sample_t:
            t_next = t_next.expand(Bq).clone()
        kw = dict(t=t, latents=x, prompt_embeds=pe, pooled_prompt_embeds=pp, guidance_scale=gs, compute_log_prob=clp,
                  return_kwargs=list(return_kwargs), noise_level=noise_level)
        if give_t_next:
            kw["t_next"] = t_next
        if replay:
            kw["next_latents"] = x_next
        if cfg_mode == "cfg":
            kw.update(negative_prompt_embeds=ne, negative_pooled_prompt_embeds=npl)
        torch.manual_seed(777 + trial)
        with torch.no_grad():
            return ad.forward(**kw)

    try:
        want, ref_error = run(None), None
    except Exception as e:          # noqa: BLE001
        want, ref_error = None, e
    if ref_error is not None:
        pytest.skip(f"the reference itself cannot serve this combination: {ref_error!r}")
    saved = (P.Engine, P.VAEDecoder, P.VAEConfig)
    P.Engine, P.VAEDecoder, P.VAEConfig = F.StandinEngine, F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c)
    try:
        got = run(P.SD3_5NativeAdapter)
    finally:
        P.Engine, P.VAEDecoder, P.VAEConfig = saved
    for k in keys:
        a, b = getattr(got, k, None), getattr(want, k, None)
        if b is None:
            assert a is None, (ctx, k, "the plugin returns a field the reference leaves out")
            continue
        assert a is not None, (ctx, k, "missing")
        assert a.dtype == b.dtype and a.shape == b.shape, (ctx, k, a.dtype, b.dtype, tuple(a.shape), tuple(b.shape))
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), (ctx, k, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("trial", range(24))
def test_differential_sweep_standalone_adapter_vs_reference_adapter(trial):
    """The STANDALONE adapter (`mi355_flow.adapter.SD3_5NativeAdapter`: the mixin on this package's own scheduler / collector / sample
    mirrors -- what bench.py and the GPU tests drive) against the reference's `SD3_5Adapter`, same sweep as above.  The engine is the
    computing double; the adapter object is assembled without its constructor (which insists on a GPU)."""
    import random
    import sys
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    from mi355_flow.adapter import SD3_5NativeAdapter
    from mi355_flow.engine import TransformerConfig
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    from mi355_flow.trajectory import compute_trajectory_indices as mirror_traj
    from oracle import make_rollout_golden as G
    rnd = random.Random(5500 + trial)
    dyn = rnd.choice(["Flow-SDE", "Dance-SDE", "CPS", "ODE"])
    storage = rnd.choice(["fp16", "bf16", None])
    N = rnd.choice([3, 4, 6, 7])
    window = sorted(rnd.sample(range(N - 1), rnd.randint(1, N - 1)))
    n_sde = rnd.randint(1, len(window))
    is_eval = rnd.random() < 0.2
    cfg_mode = rnd.choice(["cfg", "nocfg", "cfg_without_negatives"])
    gs = 1.0 if cfg_mode == "nocfg" else rnd.choice([2.0, 4.5])
    # (eval mode + log-probs: the reference collects what its expression gives at eta = 0 -- NaN under Flow- / Dance-SDE; the CPS residual
    #  is not produced by the engine and raises)
    clp = dyn != "ODE" and not (is_eval and dyn == "CPS") and rnd.random() < 0.8
    traj = rnd.choice(["all", "train"]) if clp else rnd.choice(["all", None, [0, -1], [-1], [1, 2, -2]])
    callbacks = rnd.choice([[], ["noise_level"], ["next_latents_mean"], ["noise_pred", "std_dev_t", "dt", "noise_level"]])
    explicit_gen = rnd.choice([None, None, "one", "per-sample list"])      # evaluate() passes one CPU generator per prompt
    ctx = dict(trial=trial, dyn=dyn, storage=storage, N=N, window=window, n_sde=n_sde, is_eval=is_eval, cfg=cfg_mode, gs=gs, clp=clp,
               traj=traj, callbacks=callbacks, explicit_gen=explicit_gen)
    g = torch.Generator().manual_seed(100 + trial)
    Bq = rnd.choice([1, 2, 3])
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16()     # noqa: E731
    pe, pp, ne, npl = mk(Bq, 7, 128), mk(Bq, 128), mk(Bq, 7, 128), mk(Bq, 128)

    def call(ad, traj_fn):
        ti = traj_fn(ad.scheduler.train_timesteps, N) if traj == "train" else traj
        torch.manual_seed(4242 + trial)
        kw = dict(prompt=[f"p{i}" for i in range(Bq)], prompt_ids=torch.arange(Bq * 3).reshape(Bq, 3), height=64, width=96,
                  num_inference_steps=N, guidance_scale=gs, prompt_embeds=pe, pooled_prompt_embeds=pp, compute_log_prob=clp,
                  trajectory_indices=ti, extra_call_back_kwargs=list(callbacks),
                  generator=(None if explicit_gen is None else torch.Generator().manual_seed(9 + trial) if explicit_gen == "one" else
                             [torch.Generator().manual_seed(9 + trial + 17 * i) for i in range(Bq)]))
        if cfg_mode == "cfg":
            kw.update(negative_prompt_embeds=ne, negative_pooled_prompt_embeds=npl, negative_prompt_ids=torch.zeros(Bq, 3, dtype=torch.long))
        return ad.inference(**kw)

    from flow_factory.utils.trajectory_collector import compute_trajectory_indices as ref_traj
    try:
        want = call(G.build_sd3(None, dyn, storage, window, n_sde, 0.7, is_eval, seed=trial),
                    lambda tt, n: ref_traj(train_timestep_indices=tt, num_inference_steps=n))
    except Exception as e:          # noqa: BLE001
        pytest.skip(f"the reference itself cannot serve this combination: {e!r}")
    tcfg = TransformerConfig(num_layers=1, num_heads=1, joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24, dual_layers=())
    ad = object.__new__(SD3_5NativeAdapter)
    ad.device, ad.transformer_dtype, ad._latent_storage = torch.device("cpu"), torch.bfloat16, storage
    ad.scheduler = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=window, num_sde_steps=n_sde, seed=trial, dynamics_type=dyn, shift=3.0)
    ad.engine, ad._live_weights = F.StandinEngine(tcfg), None
    ad._vae_decode, ad.vae_decoder, ad.vae_max_batch = None, None, 4
    ad.eval() if is_eval else ad.rollout()
    got = call(ad, lambda tt, n: mirror_traj(tt, n))
    for s_ in want:
        s_.image = None
    _assert_same_samples(got, want, ctx)


@pytest.mark.parametrize("family", ["flux", "qwen", "wan"])
@pytest.mark.parametrize("trial", range(10))
def test_differential_sweep_family_plugins_vs_reference_adapters(family, trial):
    """Random option combinations (dynamics, guidance / true CFG on-off, storage dtype, steps, SDE windows, trajectory selections, log-probs,
    explicit vs global generator) through the reference's `Flux1Adapter` / `QwenImageAdapter` / `Wan2_T2V_Adapter` and through the
    corresponding plugin class on its computing engine double (fused rollouts): every recorded tensor must be equal."""
    import random
    import sys
    import types
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    import mi355_flow.vae as MV
    from oracle import make_rollout_golden as G
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    rnd = random.Random(31000 + 100 * ["flux", "qwen", "wan"].index(family) + trial)
    dyn = rnd.choice(["Flow-SDE", "Dance-SDE", "CPS", "ODE"])
    storage = rnd.choice(["fp16", "bf16", None])
    N = rnd.choice([3, 4, 6])
    window = sorted(rnd.sample(range(N - 1), rnd.randint(1, N - 1)))
    n_sde = rnd.randint(1, len(window))
    eta = rnd.choice([0.5, 0.7])
    clp = dyn != "ODE" and rnd.random() < 0.7
    traj = rnd.choice(["train", "all"]) if clp else rnd.choice(["all", [-1], [0, -1]])
    gs = rnd.choice([1.0, 3.5]) if family != "flux" else rnd.choice([1.0, 3.5, 7.0])
    explicit_gen = rnd.random() < 0.5
    case = {"flux": (dyn, gs, storage, N, window, n_sde, eta), "qwen": (dyn, gs, storage, N, window, n_sde, eta),
            "wan": (dyn, gs, None, None, storage, N, window, n_sde, eta)}[family]
    run, attr, eng, names, plug = {
        "flux": (G.run_reference_flux, "FluxEngine", F.FluxStandinEngine,
                 ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias", "x_embedder.weight"], "Flux1NativeAdapter"),
        "qwen": (G.run_reference_qwen, "QwenEngine", F.QwenStandinEngine,
                 ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias"], "QwenImageNativeAdapter"),
        "wan": (G.run_reference_wan, "WanEngine", F.WanStandinEngine, ["blocks.0.attn1.to_q.weight", "blocks.0.attn1.to_q.bias"],
                "Wan2T2VNativeAdapter"),
    }[family]
    kw = dict(callbacks=False, explicit_generator=explicit_gen, traj=traj, clp=clp, seed=777 + trial)
    ctx = dict(family=family, trial=trial, case=case, **kw)
    try:
        want = run(case, **kw)
    except Exception as e:          # noqa: BLE001
        pytest.skip(f"the reference itself cannot serve this combination: {e!r}")
    saved = (getattr(P, attr), P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder)
    eng.NAMES = names
    setattr(P, attr, eng)
    P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c), F.FakeVideoVAEDecoder
    try:
        got = run(case, adapter_base=getattr(P, plug), **kw)
    finally:
        setattr(P, attr, saved[0])
        P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = saved[1:]
    assert sorted(got) == sorted(want), ctx
    for k, v in want.items():
        a, b = got[k].detach().cpu().float(), v.detach().cpu().float()
        assert a.shape == b.shape and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)), (ctx, k)


@pytest.mark.parametrize("name", sorted(WAN_CASES))
def test_wan_plugin_stepwise_and_two_expert_paths_reproduce_the_reference_adapter(name):
    """The per-step host paths of the Wan plugin -- callback capture (`extra_call_back_kwargs=['next_latents_mean']`) and the Wan2.2
    two-expert loop (expert and guidance scale per timestep, CFG decided per expert) -- against the committed reference-adapter fixture.
    Engine double: `transformer_forward` = the stand-in network of that engine's expert; the fused step kernel is replaced by the oracle
    step (bit-exact against the reference scheduler; the HIP kernel is pinned against the same fixtures on the GPU)."""
    import sys
    import types
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    import mi355_flow.vae as MV
    import mi355_flow.wan as MW
    from oracle import make_rollout_golden as G
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    saved = (P.WanEngine, MV.WanVAEDecoder, MW.sde_step)
    F.WanStandinEngineStepwise.NAMES = ["blocks.0.attn1.to_q.weight", "blocks.0.attn1.to_q.bias"]
    F.WanStandinEngineStepwise._count = 0
    P.WanEngine, MV.WanVAEDecoder, MW.sde_step = F.WanStandinEngineStepwise, F.FakeVideoVAEDecoder, F.oracle_sde_step
    try:
        live = G.run_reference_wan(name, adapter_base=P.Wan2T2VNativeAdapter)          # WITH the callback -> the step-wise path
    finally:
        P.WanEngine, MV.WanVAEDecoder, MW.sde_step = saved
    stored = _case(np.load(GOLDEN), name)
    assert sorted(live) == sorted(stored), name
    for k, v in live.items():
        assert torch.equal(v.detach().cpu().float(), stored[k].float()), (name, k)


@pytest.mark.parametrize("trial", range(24))
def test_differential_sweep_wan_forward_plugin_vs_reference_adapter(trial):
    """`forward()` of the Wan plugin (single transformer and the Wan2.2 two-expert pipeline: the expert and its guidance scale picked from
    `t` / `boundary_timestep`) against the reference's `Wan2_T2V_Adapter.forward`: scalar or (B,) integer timesteps, `t_next` given or
    derived, inferred / explicit noise level, sampling step or replay, any `return_kwargs` subset, CFG on / off."""
    import random
    import sys
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    import mi355_flow.vae as MV
    import mi355_flow.wan as MW
    from oracle import make_rollout_golden as G
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    rnd = random.Random(64000 + trial)
    dyn = rnd.choice(["Flow-SDE", "Dance-SDE", "CPS", "ODE"])
    storage = rnd.choice(["fp16", "bf16"])
    N = rnd.choice([4, 6, 8])
    window = sorted(rnd.sample(range(N - 1), rnd.randint(1, N - 1)))
    two = rnd.random() < 0.5
    gs, gs2 = rnd.choice([1.0, 4.0]), rnd.choice([None, 1.0, 3.0])
    case = (dyn, gs, gs2 if two else None, rnd.choice([0.4, 0.7]) if two else None, storage, N, window, 1, 0.7)
    step = rnd.randrange(N - 1) if dyn == "CPS" else rnd.randrange(N)
    # (`t_next` is always given, as every trainer does: deriving it from the schedule hits an index bug in the reference's UniPC step)
    per_sample_t, give_t_next = rnd.random() < 0.5, True
    noise_level = rnd.choice([None, 0.0, 0.7])
    replay = rnd.random() < 0.5
    clp = dyn != "ODE" and rnd.random() < 0.7 and (noise_level is None or noise_level > 0)
    keys = ["noise_pred", "next_latents", "next_latents_mean", "std_dev_t", "dt", "log_prob"]
    return_kwargs = rnd.sample(keys, rnd.randint(1, len(keys)))
    ctx = dict(trial=trial, case=case, step=step, per_sample_t=per_sample_t, give_t_next=give_t_next, noise_level=noise_level, replay=replay,
               clp=clp, return_kwargs=return_kwargs)
    g = torch.Generator().manual_seed(500 + trial)
    Bq = 2
    pe, ne = torch.randn(Bq, 7, G.WAN_TD, generator=g).bfloat16(), torch.randn(Bq, 7, G.WAN_TD, generator=g).bfloat16()
    sdt = {"fp16": torch.float16, "bf16": torch.bfloat16}[storage]
    x = torch.randn(Bq, 16, 2, 8, 8, generator=g).to(sdt)
    x_next = (x.float() * 0.9 + 0.1 * torch.randn(Bq, 16, 2, 8, 8, generator=g)).to(sdt)

    def run(base):
        ad, _ = G.build_wan(case, base)
        ad.scheduler.set_timesteps(N)
        ts = ad.scheduler.timesteps
        t = ts[step].expand(Bq).clone() if per_sample_t else ts[step]
        t_next = ts[step + 1] if step + 1 < N else torch.tensor(0)
        if per_sample_t:
            t_next = t_next.expand(Bq).clone()
        kw = dict(t=t, latents=x, prompt_embeds=pe, negative_prompt_embeds=ne, guidance_scale=gs, guidance_scale_2=case[2],
                  compute_log_prob=clp, return_kwargs=list(return_kwargs), noise_level=noise_level)
        if give_t_next:
            kw["t_next"] = t_next
        if replay:
            kw["next_latents"] = x_next
        torch.manual_seed(888 + trial)
        with torch.no_grad():
            return ad.forward(**kw)

    try:
        want = run(None)
    except Exception as e:          # noqa: BLE001
        pytest.skip(f"the reference itself cannot serve this combination: {e!r}")
    saved = (P.WanEngine, MV.WanVAEDecoder, MW.sde_step)
    F.WanStandinEngineStepwise.NAMES = ["blocks.0.attn1.to_q.weight", "blocks.0.attn1.to_q.bias"]
    F.WanStandinEngineStepwise._count = 0
    P.WanEngine, MV.WanVAEDecoder, MW.sde_step = F.WanStandinEngineStepwise, F.FakeVideoVAEDecoder, F.oracle_sde_step
    try:
        got = run(P.Wan2T2VNativeAdapter)
    finally:
        P.WanEngine, MV.WanVAEDecoder, MW.sde_step = saved
    for k in keys:
        a, b = getattr(got, k, None), getattr(want, k, None)
        if b is None:
            assert a is None, (ctx, k, "the plugin returns a field the reference leaves out")
            continue
        assert a is not None, (ctx, k, "missing")
        assert a.dtype == b.dtype and a.shape == b.shape, (ctx, k, a.dtype, b.dtype, tuple(a.shape), tuple(b.shape))
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), (ctx, k, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("family,name", [("flux", "flux_flow_sde_fp16"), ("flux", "flux_dance_native"), ("qwen", "qwen_flow_sde_cfg_ragged"),
                                         ("qwen", "qwen_cps_nocfg_fp16")])
@pytest.mark.parametrize("callbacks", [False, True], ids=["fused", "stepwise-callbacks"])
def test_flux_and_qwen_plugin_paths_at_the_model_level(family, name, callbacks):
    """FLUX.1 / Qwen-Image plugin rollouts -- fused and the per-step path (callback capture) -- against the reference adapters with the
    stand-in placed INSIDE the model boundary (`oracle.standin.*_transformer_call`: diffusers' own first arithmetic on the adapter's
    `timestep` / `guidance`, then the stand-in), so that the engine double receives the values the PRODUCT's host code computes for the
    network (`t_model` = round_storage(t / 1000) * 1000, guidance likewise; Qwen: `model_timestep`).  Per-step paths run on the CPU
    stand-in of the fused step kernel."""
    import sys
    import types
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    import mi355_flow.flux as MF
    import mi355_flow.qwen as MQ
    import mi355_flow.vae as MV
    from oracle import make_rollout_golden as G
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    run, attr, eng, names, plug = {
        "flux": (G.run_reference_flux, "FluxEngine", F.FluxStandinEngineModel,
                 ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias", "x_embedder.weight"], "Flux1NativeAdapter"),
        "qwen": (G.run_reference_qwen, "QwenEngine", F.QwenStandinEngineModel,
                 ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias"], "QwenImageNativeAdapter"),
    }[family]
    want = run(name, callbacks=callbacks, model_level=True)
    saved = (getattr(P, attr), P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder, MF.sde_step, MQ.sde_step)
    eng.NAMES = names
    setattr(P, attr, eng)
    P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c), F.FakeVideoVAEDecoder
    MF.sde_step = MQ.sde_step = F.oracle_sde_step
    try:
        got = run(name, adapter_base=getattr(P, plug), callbacks=callbacks, model_level=True)
    finally:
        setattr(P, attr, saved[0])
        P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder, MF.sde_step, MQ.sde_step = saved[1:]
    assert sorted(got) == sorted(want), name
    for k, v in want.items():
        assert torch.equal(got[k].detach().cpu().float(), v.detach().cpu().float()), (name, k)


@pytest.mark.parametrize("family", ["flux", "qwen"])
@pytest.mark.parametrize("trial", range(16))
def test_differential_sweep_flux_qwen_forward_plugin_vs_reference_adapter(family, trial):
    """`forward()` of the FLUX.1 / Qwen-Image plugins against the reference adapters' `forward` (stand-in inside the model boundary, CPU
    stand-in of the fused step kernel): scalar or (B,) timesteps on and off the schedule grid's storage rounding, `t_next` given or
    derived, inferred / explicit noise level, sampling step or replay, `return_kwargs` subsets, guidance values; Qwen: ragged prompt lists,
    true CFG on / off."""
    import random
    import sys
    import types
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    import mi355_flow.flux as MF
    import mi355_flow.qwen as MQ
    import mi355_flow.vae as MV
    from oracle import make_rollout_golden as G
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    rnd = random.Random(81000 + 100 * (family == "qwen") + trial)
    dyn = rnd.choice(["Flow-SDE", "Dance-SDE", "CPS", "ODE"])
    storage = rnd.choice(["fp16", "bf16"])
    N = rnd.choice([4, 6, 8])
    window = sorted(rnd.sample(range(N - 1), rnd.randint(1, N - 1)))
    gs = rnd.choice([1.0, 3.5, 7.0]) if family == "flux" else rnd.choice([1.0, 4.0])
    case = (dyn, gs, storage, N, window, 1, 0.7)
    step = rnd.randrange(N - 1) if dyn == "CPS" else rnd.randrange(N)
    per_sample_t, give_t_next = rnd.random() < 0.5, rnd.random() < 0.6
    noise_level = rnd.choice([None, 0.0, 0.7])
    replay = rnd.random() < 0.5
    clp = dyn != "ODE" and rnd.random() < 0.7 and (noise_level is None or noise_level > 0)
    keys = ["noise_pred", "next_latents", "next_latents_mean", "std_dev_t", "dt", "log_prob"]
    return_kwargs = rnd.sample(keys, rnd.randint(1, len(keys)))
    ctx = dict(family=family, trial=trial, case=case, step=step, per_sample_t=per_sample_t, give_t_next=give_t_next, noise_level=noise_level,
               replay=replay, clp=clp, return_kwargs=return_kwargs)
    g = torch.Generator().manual_seed(600 + trial)
    Bq, hp, wp = 2, 4, 6
    sdt = {"fp16": torch.float16, "bf16": torch.bfloat16}[storage]
    x = torch.randn(Bq, hp * wp, 64, generator=g).to(sdt)
    x_next = (x.float() * 0.9 + 0.1 * torch.randn(Bq, hp * wp, 64, generator=g)).to(sdt)
    if family == "flux":
        from oracle import flux_ref as FR
        extra = dict(prompt_embeds=torch.randn(Bq, 7, 128, generator=g).bfloat16(), pooled_prompt_embeds=torch.randn(Bq, 128, generator=g).bfloat16(),
                     img_ids=FR.prepare_img_ids(hp, wp).to(sdt))
        build, attr, eng, names, plug = (G.build_flux, "FluxEngine", F.FluxStandinEngineModel,
                                         ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias", "x_embedder.weight"],
                                         "Flux1NativeAdapter")
    else:
        lens, nlens = [5, 9], [3, 4]
        extra = dict(prompt_embeds=[torch.randn(n, G.QJ, generator=g).bfloat16() for n in lens],
                     prompt_embeds_mask=[torch.ones(n, dtype=torch.long) for n in lens], img_shapes=[[(1, hp, wp)]] * Bq)
        if gs > 1:
            extra.update(negative_prompt_embeds=[torch.randn(n, G.QJ, generator=g).bfloat16() for n in nlens],
                         negative_prompt_embeds_mask=[torch.ones(n, dtype=torch.long) for n in nlens])
        build, attr, eng, names, plug = (G.build_qwen, "QwenEngine", F.QwenStandinEngineModel,
                                         ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias"], "QwenImageNativeAdapter")

    def run(base):
        from flow_factory.scheduler import set_scheduler_timesteps
        ad = build(case, base, model_level=True)
        ts = set_scheduler_timesteps(scheduler=ad.scheduler, num_inference_steps=N, seq_len=hp * wp, device=torch.device("cpu"))
        t = ts[step].expand(Bq).clone() if per_sample_t else ts[step]
        t_next = ts[step + 1] if step + 1 < N else torch.tensor(0.0)
        if per_sample_t:
            t_next = t_next.expand(Bq).clone()
        kw = dict(t=t, latents=x, guidance_scale=gs, compute_log_prob=clp, return_kwargs=list(return_kwargs), noise_level=noise_level, **extra)
        if give_t_next:
            kw["t_next"] = t_next
        if replay:
            kw["next_latents"] = x_next
        torch.manual_seed(999 + trial)
        with torch.no_grad():
            return ad.forward(**kw)

    try:
        want = run(None)
    except Exception as e:          # noqa: BLE001
        pytest.skip(f"the reference itself cannot serve this combination: {e!r}")
    saved = (getattr(P, attr), P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder, MF.sde_step, MQ.sde_step)
    eng.NAMES = names
    setattr(P, attr, eng)
    P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder = F.FakeVAEDecoder, types.SimpleNamespace(from_hf=lambda c: c), F.FakeVideoVAEDecoder
    MF.sde_step = MQ.sde_step = F.oracle_sde_step
    try:
        got = run(getattr(P, plug))
    finally:
        setattr(P, attr, saved[0])
        P.VAEDecoder, P.VAEConfig, MV.WanVAEDecoder, MF.sde_step, MQ.sde_step = saved[1:]
    for k in keys:
        a, b = getattr(got, k, None), getattr(want, k, None)
        if b is None:
            assert a is None, (ctx, k, "the plugin returns a field the reference leaves out")
            continue
        assert a is not None, (ctx, k, "missing")
        assert a.dtype == b.dtype and a.shape == b.shape, (ctx, k, a.dtype, b.dtype, tuple(a.shape), tuple(b.shape))
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), (ctx, k, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("case", [("Flow-SDE", 5.0, None, None, "fp16", 6, [1, 2, 3], 2, 0.7),          # Wan2.1, CFG, fp16 storage
                                  ("Flow-SDE", 1.0, None, None, "bf16", 3, [0, 1], 1, 0.7),              # no CFG, 3 steps (warm-up + final-order-1 only)
                                  ("CPS", 4.0, 3.0, 0.6, "bf16", 7, [0, 1, 2, 3], 2, 0.8),                # Wan2.2 two experts, a guidance scale each
                                  ("ODE", 4.0, 1.0, 0.5, None, 5, [0, 1], 1, 0.7)],                       # second expert without CFG, fp32 storage
                         ids=["wan21_cfg_fp16", "nocfg_3_steps", "wan22_two_expert", "wan22_low_expert_nocfg_fp32"])
def test_wan_plugin_evaluation_mode_sampler_follows_the_reference_adapter_loop(case):
    """Evaluation-mode sampling (round 5: native; reference wan2_t2v.py:346-375 with `scheduler.step` in its `is_eval` branch,
    unipc_multistep.py:282-285 = diffusers' UniPC multistep predictor-corrector).  The reference's own `Wan2_T2V_Adapter.inference`, in eval
    mode, on the diffusers stub whose solver step is oracle/unipc_ref.py (the published algorithm tensor by tensor) against the plugin's
    `_rollout_eval`: per-step engine forwards on the stand-in network double + the solver as host-side linear coefficients
    (mi355_flow/unipc.py; its two HIP kernels replaced here by torch statements of what they compute).  Pins the control flow -- one latent
    draw and no step noise, the expert / guidance / CFG decision per step, cast_latents after every step, corrector from the second step on,
    warm-up and final order -- with the solver body itself unpinned on both sides (diffusers is not in this image)."""
    import sys
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    ref_package.install()
    sys.path.insert(0, os.path.dirname(__file__))
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    import mi355_flow.unipc as MU
    import mi355_flow.vae as MV
    import mi355_flow.wan as MW
    from oracle import make_rollout_golden as G
    from oracle import rollout_ref as R
    if P._RefAdapter is None:
        import importlib
        P = importlib.reload(P)
    kw = dict(callbacks=False, traj="all", clp=False, seed=4242, evaluation=True)
    want = G.run_reference_wan(case, **kw)

    def convert(v_text, v_uncond, guidance, sample, sigma):          # mi355_unipc_convert
        v = v_text if v_uncond is None else R.cfg_combine_bf16(v_uncond, v_text, float(guidance))
        return sample.float() - (torch.tensor(float(sigma), dtype=torch.float32) * v.float()).to(v.dtype).float()

    def lincomb(tensors, coefs, out_dtype):                          # mi355_op_lincomb
        acc = None
        for t, c in zip(tensors, coefs):
            term = (torch.tensor(float(c), dtype=torch.float32) * t.float()).to(t.dtype).float()
            acc = term if acc is None else acc + term
        return acc.to(out_dtype)

    saved = (P.WanEngine, MV.WanVAEDecoder, MW.sde_step, MU.unipc_convert, MU.lincomb)
    F.WanStandinEngineStepwise.NAMES = ["blocks.0.attn1.to_q.weight", "blocks.0.attn1.to_q.bias"]
    F.WanStandinEngineStepwise._count = 0
    P.WanEngine, MV.WanVAEDecoder, MW.sde_step, MU.unipc_convert, MU.lincomb = F.WanStandinEngineStepwise, F.FakeVideoVAEDecoder, F.oracle_sde_step, convert, lincomb
    calls = []
    try:
        orig = F.WanStandinPlanStepwise.transformer_forward

        def spy(self, latents, t, enc_a, enc_b=None):
            calls.append((self.engine_expert, 1 if enc_b is None else 2))
            return orig(self, latents, t, enc_a, enc_b)
        F.WanStandinPlanStepwise.transformer_forward = spy
        got = G.run_reference_wan(case, adapter_base=P.Wan2T2VNativeAdapter, **kw)
    finally:
        F.WanStandinPlanStepwise.transformer_forward = orig
        P.WanEngine, MV.WanVAEDecoder, MW.sde_step, MU.unipc_convert, MU.lincomb = saved
    N = case[5]
    assert len(calls) == N                                            # one engine forward per step (both CFG branches in one batch), nothing else
    assert sorted(got) == sorted(want), (sorted(got), sorted(want))
    for k in ("timesteps", "sigmas", "latent_index_map", "latents_dtype"):
        assert torch.equal(got[k].float(), want[k].float()), k
    a, b = got["all_latents"], want["all_latents"]
    assert a.shape == b.shape and a.shape[1] == N + 1
    assert torch.equal(a[:, 0], b[:, 0])                              # the initial latents: same draw, same cast
    # Close, not equal: the two statements of the solver round different intermediate sums, and the stand-in network quantises its input and
    # output to bf16 -- a last-bit difference in a latent flips a bf16 rounding and comes back as a bf16 ulp of the prediction (measured: up to
    # 4e-3 of the latent range after four steps).  A control-flow error (wrong expert or guidance, a missing corrector, an extra noise draw)
    # moves the latents by O(0.1 ... 1) of their range.  (bf16 storage: every stored latent is itself rounded to 2^-8.)
    tol = 5e-2 if case[4] == "bf16" else 2e-2
    for i in range(1, N + 1):
        d = float((a[:, i] - b[:, i]).abs().max())
        assert d <= tol * max(1.0, float(b[:, i].abs().max())), (i, d)
    assert "log_probs" not in got and "log_probs" not in want
