"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz.

Run in the build container (needs /root/reference):  python -m oracle.make_golden

Fixtures marked [REF] are outputs of the reference's OWN code executed under
oracle/diffusers_stub.py (see oracle/ref_loader.py); they pin oracle/scheduler_ref.py,
oracle/advantage_ref.py and the product's host-side mirrors.  Fixtures marked [SELF]
come from oracle/mmditx_ref.py (parity unpinned: the denoiser lives in un-vendored
diffusers) and only guard the oracle against silent drift.
"""
from __future__ import annotations

import os
import sys
import threading
import types

import numpy as np
import torch

from . import advantage_ref, mmditx_ref, ref_loader, rollout_ref, scheduler_ref

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def _np(t):
    if t is None:
        return np.zeros((0,), np.float32)
    return t.detach().float().cpu().numpy()


# ------------------------------------------------------------------ [REF] scheduler step KATs
def gen_scheduler_steps(ns):
    out = {}
    cases = []
    N = 10
    for dyn in ("Flow-SDE", "Dance-SDE", "CPS", "ODE"):
        for sd_name, sdt in DT.items():
            # (step index, eta, compute_log_prob)
            for (i, eta, clp) in ((0, 0.7, True), (3, 0.7, True), (N - 1, 0.8, True), (2, 0.0, False)):
                if dyn == "Dance-SDE" and i == N - 1:
                    pass  # sigma_next = 0 is fine for Dance-SDE (sqrt(-dt) > 0)
                if dyn == "CPS" and i == N - 1:
                    continue  # std_dev_t = sigma_next*sin = 0 -> log-prob of a delta; the reference never trains there
                cases.append((dyn, sd_name, i, eta, clp))
    for ci, (dyn, sd_name, i, eta, clp) in enumerate(cases):
        sched = ref_loader.make_reference_scheduler(ns, noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1,
                                                    seed=42, dynamics_type=dyn)
        ts = ns.set_scheduler_timesteps(sched, N, seq_len=256)
        g = torch.Generator().manual_seed(1000 + ci)
        B, C, H, W = 2, 16, 4, 4
        lat = torch.randn(B, C, H, W, generator=g).to(DT[sd_name])
        v = torch.randn(B, C, H, W, generator=g).to(torch.bfloat16)
        t = ts[i]
        t_next = ts[i + 1] if i + 1 < N else torch.tensor(0.0)
        gn = torch.Generator().manual_seed(5000 + ci)
        o = sched.step(noise_pred=v, timestep=t, latents=lat, timestep_next=t_next, generator=gn,
                       noise_level=eta, compute_log_prob=clp)
        gn2 = torch.Generator().manual_seed(5000 + ci)
        eps = torch.randn(B, C, H, W, generator=gn2, dtype=torch.float32)
        k = f"c{ci}"
        out[k + "_meta"] = np.array([dyn, sd_name, str(i), repr(eta), str(int(clp)), repr(float(t)), repr(float(t_next)),
                                     repr(float(sched.sigmas[1]))])
        out[k + "_latents"] = _np(lat)
        out[k + "_noise_pred"] = _np(v)
        out[k + "_eps"] = _np(eps)
        out[k + "_next"] = _np(o.next_latents)
        out[k + "_mean"] = _np(o.next_latents_mean)
        out[k + "_std"] = _np(o.std_dev_t)
        out[k + "_dt"] = _np(o.dt)
        out[k + "_logp"] = _np(o.log_prob)
        if clp and dyn != "ODE":
            # replay on the stored (storage-dtype) next latents: grpo.py:229-263
            nxt = o.next_latents.to(DT[sd_name])
            o2 = sched.step(noise_pred=v, timestep=t, latents=lat, timestep_next=t_next, next_latents=nxt,
                            noise_level=eta, compute_log_prob=True)
            out[k + "_replay_logp"] = _np(o2.log_prob)
    out["num_cases"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(OUT, "scheduler_steps.npz"), **out)
    return len(cases)


# ------------------------------------------------------------------ [REF] schedule + SDE-step selection
def gen_schedule(ns):
    out = {}
    for N in (4, 10, 28):
        sched = ref_loader.make_reference_scheduler(ns)
        ts = ns.set_scheduler_timesteps(sched, N, seq_len=4096)
        out[f"static_N{N}_timesteps"] = _np(ts)
        out[f"static_N{N}_sigmas"] = _np(sched.sigmas)
        for seq in (256, 1024, 4096):
            sched = ref_loader.make_reference_scheduler(ns, use_dynamic_shifting=True, shift=1.0)
            ts = ns.set_scheduler_timesteps(sched, N, seq_len=seq)
            out[f"dyn_N{N}_S{seq}_timesteps"] = _np(ts)
            out[f"dyn_N{N}_S{seq}_sigmas"] = _np(sched.sigmas)
    sel = []
    configs = [([1, 2, 3], 1), ([1, 2, 3, 4, 5], 2), ([0, 1, 2, 3, 4], 3), (None, None), ([2, 5, 7], 5)]
    for ci, (steps, n) in enumerate(configs):
        for seed in range(0, 24):
            sched = ref_loader.make_reference_scheduler(ns, noise_level=0.7, sde_steps=steps, num_sde_steps=n, seed=seed)
            ns.set_scheduler_timesteps(sched, 10, seq_len=256)
            cur = sched.current_sde_steps.tolist()
            nl = sched.get_noise_levels().tolist()
            sel.append((ci, seed, cur, nl))
            # noise level looked up by timestep value (sd3_5.py:274)
            for i in (0, 3, 9):
                assert abs(sched.get_noise_level_for_timestep(sched.timesteps[i]) - nl[i]) < 1e-6
    out["select_cfg"] = np.array([repr(c) for c in configs])
    out["select_rows"] = np.array([repr(r) for r in sel])
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **out)


# ------------------------------------------------------------------ [REF] collectors
def gen_collectors(ns):
    rows = []
    for N in (4, 10, 28):
        for train in ([1], [1, 3], [0, 1, 2], [N - 1], []):
            train = [t for t in train if t < N]
            idx = ns.compute_trajectory_indices(torch.tensor(train, dtype=torch.int64), N)
            for spec in (idx, "all", None, [0, -1]):
                lc = ns.TrajectoryCollector(spec, N)
                pc = ns.TrajectoryCollector(spec, N)
                lc.collect(torch.zeros(1), 0)
                for i in range(N):
                    lc.collect(torch.full((1,), float(i + 1)), i + 1)
                    if i in train:
                        pc.collect(torch.full((1,), float(i)), i)
                lm, pm = lc.get_index_map(), pc.get_index_map()
                rows.append(repr(dict(N=N, train=train, spec=spec if spec is not None else "None",
                                      traj_idx=idx,
                                      latent_map=None if lm is None else lm.tolist(),
                                      logp_map=None if pm is None else pm.tolist(),
                                      lat_vals=None if lc.get_result() is None else [float(v) for v in lc.get_result()],
                                      logp_vals=None if pc.get_result() is None else [float(v) for v in pc.get_result()])))
    np.savez_compressed(os.path.join(OUT, "collectors.npz"), rows=np.array(rows))


# ------------------------------------------------------------------ [REF] advantages
class _FakeSample:
    def __init__(self, uid):
        self.unique_id = uid
        self.extra_kwargs = {}


class _FakeAccel:
    """Single-node stand-in for accelerate.Accelerator: `reduce(sum)` over threads."""

    def __init__(self, rank, world, hub):
        self.process_index, self.num_processes, self.hub = rank, world, hub
        self.device = torch.device("cpu")

    def reduce(self, t, reduction="sum"):
        assert reduction == "sum"
        if self.num_processes == 1:
            return t
        self.hub["slots"][self.process_index] = t.clone()
        self.hub["barrier"].wait()
        tot = sum(self.hub["slots"])
        self.hub["barrier"].wait()
        return tot

    def gather(self, t):
        if self.num_processes == 1:
            return t
        self.hub["slots"][self.process_index] = t.clone()
        self.hub["barrier"].wait()
        tot = torch.cat(list(self.hub["slots"]), 0)
        self.hub["barrier"].wait()
        return tot


def _load_reference_advantage():
    import importlib.util

    ref_loader.load()
    for name, attrs in (
        ("flow_factory.samples", dict(BaseSample=_FakeSample)),
        ("flow_factory.rewards", dict(RewardProcessor=type("RewardProcessor", (), {}))),
        ("flow_factory.utils.dist", dict(global_zero_std_ratio=lambda *a, **k: 0.0,
                                         global_tensor_stats_batch=lambda *a, **k: {})),
    ):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    pkg = types.ModuleType("flow_factory.advantage")
    pkg.__path__ = [os.path.join(ref_loader.REF_PKG, "advantage")]
    sys.modules.setdefault("flow_factory.advantage", pkg)
    spec = importlib.util.spec_from_file_location(
        "flow_factory.advantage.advantage_processor",
        os.path.join(ref_loader.REF_PKG, "advantage", "advantage_processor.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    AP = mod.AdvantageProcessor
    AP._build_weighted_sum_log_data = lambda self, *a, **k: {}
    AP._build_gdpo_log_data = lambda self, *a, **k: {}
    return AP


def gen_advantages():
    AP = _load_reference_advantage()
    rng = np.random.default_rng(7)
    M_, K = 8, 4
    ids = np.repeat(rng.integers(1, 2**40, size=M_), K)
    rewards = {"clip": rng.normal(0.3, 0.1, M_ * K).astype(np.float32),
               "pick": rng.normal(20.0, 2.0, M_ * K).astype(np.float32)}
    rewards["clip"][4:8] = 0.25  # a zero-variance group exercises the 1e-6 floor
    weights = {"clip": 1.0, "pick": 0.5}
    out = dict(ids=ids, clip=rewards["clip"], pick=rewards["pick"], K=np.array([K]))
    for world in (1, 2):
        hub = dict(slots=[None] * world, barrier=threading.Barrier(world))
        for mode, kw in (("sum_gstd", dict(global_std=True)), ("sum_lstd", dict(global_std=False)), ("gdpo", {})):
            res = [None] * world

            def run(rank):
                ap = AP(_FakeAccel(rank, world, hub), weights, K, sampler_type="group_contiguous",
                        verbose=False, **(kw if mode != "gdpo" else {}))
                per = M_ * K // world
                sl = slice(rank * per, (rank + 1) * per)
                samples = [_FakeSample(int(u)) for u in ids[sl]]
                rw = {k: torch.from_numpy(v[sl]) for k, v in rewards.items()}
                fn = ap.compute_gdpo if mode == "gdpo" else ap.compute_weighted_sum
                res[rank] = fn(samples, rw, store_to_samples=False).numpy()

            th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
            [t.start() for t in th]
            [t.join() for t in th]
            out[f"{mode}_w{world}"] = np.concatenate(res)
    np.savez_compressed(os.path.join(OUT, "advantages.npz"), **out)


# ------------------------------------------------------------------ [SELF] tiny denoiser + rollout
def gen_mmdit_tiny():
    cfg = mmditx_ref.tiny_config()
    sd = mmditx_ref.make_synthetic_state_dict(cfg, seed=1234, std=0.08)
    g = torch.Generator().manual_seed(4321)
    B, Nt = 2, 13
    x = torch.randn(B, 16, 8, 8, generator=g)
    e = torch.randn(B, Nt, cfg.joint_attention_dim, generator=g)
    p = torch.randn(B, cfg.pooled_projection_dim, generator=g)
    t = torch.tensor([900.0, 500.0])
    y = mmditx_ref.mmdit_forward(sd, cfg, x, t, e, p)
    ts, sig = scheduler_ref.make_schedule(4, shift=3.0)
    init, noise = rollout_ref.draw_rollout_noise(B, 16, 8, 8, 4, torch.bfloat16, torch.Generator().manual_seed(42))
    nl = scheduler_ref.noise_levels(4, scheduler_ref.current_sde_steps([1, 2, 3], 1, 42, 4), 0.7)
    ro = rollout_ref.rollout(sd, cfg, e, p, None, None, 1.0, init, noise, ts, sig, nl.tolist(), torch.float16)
    np.savez_compressed(os.path.join(OUT, "mmdit_tiny_self.npz"), y=_np(y), all_latents=_np(ro["all_latents"]),
                        log_probs=_np(ro["log_probs"]), noise_levels=_np(nl))


def gen_vae_tiny():
    """[SELF] VAE decoder restatement (parity unpinned, oracle/vae_ref.py): drift guard only."""
    from oracle import vae_ref
    cfg = vae_ref.tiny_config()
    sd = vae_ref.make_synthetic_state_dict(cfg, seed=99)
    z = torch.randn(1, 16, 4, 4, generator=torch.Generator().manual_seed(98))
    raw = vae_ref.vae_decode(sd, cfg, z, postprocess=False)
    img = vae_ref.vae_decode(sd, cfg, z, quant=lambda t: t.bfloat16().float(), postprocess=True)
    np.savez_compressed(os.path.join(OUT, "vae_tiny_self.npz"), raw=_np(raw), img_bf16=_np(img))


def gen_flux_tiny():
    """[SELF] FLUX.1 transformer restatement (parity unpinned, oracle/flux_ref.py): drift guard only."""
    from oracle import flux_ref as Fx
    cfg = Fx.tiny_config()
    sd = Fx.make_synthetic_state_dict(cfg, seed=31)
    g = torch.Generator().manual_seed(32)
    B, h, w, Nt = 2, 4, 6, 5
    x = Fx.pack_latents(torch.randn(B, 16, h, w, generator=g))
    enc = torch.randn(B, Nt, cfg.joint_attention_dim, generator=g)
    pool = torch.randn(B, cfg.pooled_projection_dim, generator=g)
    y = Fx.flux_forward(sd, cfg, x, torch.tensor([0.9, 0.4]), torch.tensor([3.5, 3.5]), pool, enc, Fx.prepare_img_ids(h // 2, w // 2))
    np.savez_compressed(os.path.join(OUT, "flux_tiny_self.npz"), y=_np(y))


def gen_wan_tiny():
    """[SELF] Wan2.1 transformer restatement (parity unpinned, oracle/wan_ref.py): drift guard only."""
    from oracle import wan_ref as W
    cfg = W.tiny_config()
    sd = W.make_synthetic_state_dict(cfg, seed=41)
    g = torch.Generator().manual_seed(42)
    B, T, h, w, Nt = 2, 2, 4, 6, 5
    x = torch.randn(B, 16, T, h, w, generator=g)
    enc = torch.randn(B, Nt, cfg.text_dim, generator=g)
    y = W.wan_forward(sd, cfg, x, torch.tensor([874.0, 249.0]), enc)
    np.savez_compressed(os.path.join(OUT, "wan_tiny_self.npz"), y=_np(y))


# ------------------------------------------------------------------ [REF] group-contiguous sampler (DP partitioner)
def gen_sampler():
    import importlib.util

    ref_loader.load()
    pkg = types.ModuleType("flow_factory.data_utils")
    pkg.__path__ = [os.path.join(ref_loader.REF_PKG, "data_utils")]
    sys.modules.setdefault("flow_factory.data_utils", pkg)
    spec = importlib.util.spec_from_file_location("flow_factory.data_utils.sampler",
                                                  os.path.join(ref_loader.REF_PKG, "data_utils", "sampler.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    rows = []
    for (ds, bs, k, m, world, seed) in ((100, 2, 4, 8, 2, 42), (64, 4, 16, 48, 8, 7), (10, 1, 3, 4, 1, 0)):
        for rank in range(world):
            smp = mod.GroupContiguousSampler(list(range(ds)), bs, k, m, world, rank, seed)
            it = iter(smp)
            batches = [next(it) for _ in range(2 * smp.num_batches_per_epoch)]  # two epochs
            rows.append(repr(dict(ds=ds, bs=bs, k=k, m=m, world=world, seed=seed, rank=rank,
                                  nb=smp.num_batches_per_epoch, batches=batches)))
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), rows=np.array(rows))


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = ref_loader.load()
    n = gen_scheduler_steps(ns)
    gen_schedule(ns)
    gen_collectors(ns)
    gen_advantages()
    gen_mmdit_tiny()
    gen_vae_tiny()
    gen_flux_tiny()
    gen_wan_tiny()
    gen_sampler()
    print(f"wrote fixtures to {OUT} ({n} scheduler step cases)")


if __name__ == "__main__":
    main()
