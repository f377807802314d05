"""world_size-2 gloo tests (CPU) of the N>1 path: the rollout shards by whole prompt groups with no
data-path collective (GroupContiguousSampler); the only collective is the (n, sum, sum_sq)
all-reduce behind the global advantage std.  Sharded results must equal the reference's 2-rank
fixtures and the single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, PKG, ROOT


def _worker(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mi355_flow import advantage as adv
    from mi355_flow import sampler as smp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(GOLDEN, "advantages.npz"))
    n = len(z["ids"]) // world
    sl = slice(rank * n, (rank + 1) * n)
    rewards = {"clip": z["clip"][sl], "pick": z["pick"][sl]}
    w = {"clip": 1.0, "pick": 0.5}
    K = int(z["K"][0])
    res = dict(
        sum_gstd=adv.compute_weighted_sum(rewards, w, z["ids"][sl], K, True).numpy(),
        sum_lstd=adv.compute_weighted_sum(rewards, w, z["ids"][sl], K, False).numpy(),
        gdpo=adv.compute_gdpo(rewards, w, z["ids"][sl]).numpy(),
    )
    # DP partition: ranks own disjoint whole groups
    s = smp.GroupContiguousSampler(64, 4, 4, 16, world, rank, seed=3)
    res["owned"] = np.array(sorted(set(i for b in s.epoch_batches(0) for i in b)))
    # whole-job throughput reduction used by bench.py: MAX over ranks of the timed interval
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["tmax"] = t.numpy()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_advantages_and_partition(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(GOLDEN, "advantages.npz"))
    r = [np.load(os.path.join(tmp_path, f"r{i}.npz")) for i in range(world)]
    for key, gold in (("sum_gstd", "sum_gstd_w2"), ("sum_lstd", "sum_lstd_w2"), ("gdpo", "gdpo_w2")):
        got = np.concatenate([x[key] for x in r])
        np.testing.assert_allclose(got, z[gold], rtol=2e-6, atol=2e-6)           # == reference, 2 ranks
        np.testing.assert_allclose(got, z[gold.replace("_w2", "_w1")], rtol=1e-5, atol=1e-5)  # == unsharded
    owned = [set(x["owned"].tolist()) for x in r]
    assert not (owned[0] & owned[1]) and len(owned[0]) == len(owned[1]) == 8
    assert all(float(x["tmax"][0]) == 2.0 for x in r)


def test_bench_self_launches_n_ranks_and_aggregates():
    """`python bench.py --gpus 2` typed by hand re-executes itself under torch.distributed.run.  GPU-less check of
    the launcher path: gloo rendezvous on 127.0.0.1, barrier-bracketed timing, MAX over ranks, per-rank gather, ONE JSON line from
    rank 0 -- with the rollout replaced by a sleep (`--dry-run`; the engine itself is covered by the -m gpu tests)."""
    import json
    import subprocess
    import sys

    from conftest import ROOT
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run",
                        "--batch", "4", "--denoise-steps", "10"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["data"] == "dry-run" and j["scaling"] == "weak"
    assert j["world"]["world_size"] == 2 and len(j["world"]["per_rank_denoise_steps_per_s"]) == 2
    wall = j["ms_per_step"] * 1e-3 * 2
    assert wall >= 0.05 * 2 * 2 * 0.99                    # the slow rank (2 x 0.1 s) bounds the job: MAX over ranks
    assert abs(j["value"] - 4 * 10 * 2 * 2 / wall) < 0.02 * j["value"]      # whole-job aggregate over both ranks
    # the policy-update leg: one optimize() micro-step under DistributedDataParallel, synchronising vs no_sync(), MAX over
    # ranks, bytes handed to the collective -- assembled here on gloo around a stand-in module; the real path wraps the SD3.5 module bound to
    # the engine (bench.py: `optimize_step_ddp`, world > 1 only)
    leg = j["optimize_step_ddp"]
    assert leg["world_size"] == 2 and leg["backend"] == "gloo" and len(leg["per_rank_ms"]) == 2 and all(len(r) == 2 for r in leg["per_rank_ms"])
    assert leg["bytes_reduced_per_microstep"] == 4 * (64 * 128 + 128 + 128 * 64 + 64) and leg["trainable_tensors"] == 4
    assert leg["ms_per_microstep_allreduce"] > 0 and leg["ms_per_microstep_no_sync"] > 0
    assert abs(leg["ms_exposed_allreduce"] - (leg["ms_per_microstep_allreduce"] - leg["ms_per_microstep_no_sync"])) < 2e-3
    assert leg["ms_per_microstep_allreduce"] == max(r[0] for r in leg["per_rank_ms"])


def test_bench_single_rank_line_has_no_ddp_leg():
    """N = 1: no process group, no collective -- the leg must not appear (and `--no-ddp-step` removes it at N > 1)."""
    import json
    import subprocess
    import sys

    from conftest import ROOT
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "optimize_step_ddp" not in json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run", "--no-ddp-step"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "optimize_step_ddp" not in json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])


def _ddp_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mi355_flow import autograd as AG
    torch.manual_seed(0)
    mod = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    mod[0].bias.requires_grad_(False)                       # a frozen parameter: not in the reducer
    ddp = torch.nn.parallel.DistributedDataParallel(mod)
    params = [p for p in mod.parameters() if p.requires_grad]

    class EngineLike(torch.autograd.Function):              # gradients come from outside torch, like _DenoiseReplayFn's
        @staticmethod
        def forward(ctx, scale, *ws):
            ctx.scale, ctx.shapes = scale, [w.shape for w in ws]
            return torch.zeros(())

        @staticmethod
        def backward(ctx, g):
            return (None,) + tuple(torch.full(s, float(ctx.scale)) * g for s in ctx.shapes)

    found = AG._ddp_of(ddp)
    assert found is ddp
    found._pre_forward()
    out = EngineLike.apply(float(rank + 1), *params)
    found._post_forward(out)
    out.backward()
    ok = all(torch.allclose(p.grad, torch.full_like(p.grad, (1.0 + 2.0) / 2)) for p in params) and mod[0].bias.grad is None
    # gradient accumulation: no all-reduce inside no_sync()
    for p in params:
        p.grad = None
    with ddp.no_sync():
        found._pre_forward()
        out = EngineLike.apply(float(rank + 1), *params)
        found._post_forward(out)
        out.backward()
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p.grad, float(rank + 1))) for p in params)
    np.save(os.path.join(out_dir, f"ddp_{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_reducer_is_armed_around_the_engine_autograd_node(tmp_path):
    """north star: "gradient all-reduce on RCCL for the policy update".  The engine's autograd node bypasses `module.forward`, so
    `mi355_flow.autograd.denoise_replay` arms DDP's reducer itself (`_pre_forward` / `_post_forward`, what DDP.forward does): the
    gradients the node returns are then bucket-all-reduced (here on gloo; `nccl` = RCCL on the GPUs) exactly like ordinary ones, and
    `no_sync()` accumulation windows (accelerator.accumulate, trainers/grpo.py:236) skip the reduction."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(int(np.load(tmp_path / f"ddp_{r}.npy")[0]) == 1 for r in range(2))


def _qwen_replay_ddp_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    import _plugin_fakes as F
    import mi355_flow.engine as ME
    from mi355_flow import autograd as AG
    from mi355_flow.binding import LiveWeights
    ME.sde_step, ME.sde_step_bwd = F.oracle_sde_step, F.oracle_sde_step_bwd          # the fused step kernels' CPU stand-ins (oracle arithmetic)
    names = ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.to_q.bias", "transformer_blocks.0.attn.to_k.weight", "img_in.weight"]
    mod = F.build_module_tree({n: ((8, 8) if n.endswith("weight") else (8,)) for n in names}, buffers=(), seed=3)      # same values on both ranks
    mod.get_submodule("img_in").weight.requires_grad_(False)                      # frozen: outside the reducer and outside the native scope
    ddp = torch.nn.parallel.DistributedDataParallel(mod)

    class Eng:                                                                    # host API of QwenEngine that the replay node uses
        def __init__(self):
            self.bound, self.bufs = {}, {}

        def param_names(self):
            return list(names)

        def bind_tensor(self, name, t):
            self.bound[name] = t.detach().float().clone()

        def finish_binding(self):
            pass

        def grad_supported(self, name):
            return 1 if ".attn.to_" in name else 0

        def set_grad(self, name, t):
            assert self.grad_supported(name)
            self.bufs[name] = t

        def clear_grads(self):
            self.bufs = {}

    class Plan:                                                                   # QwenPlan.forward_train / backward
        def __init__(self, eng):
            self.engine, self._train_serial = eng, 0

        def forward_train(self, latents, t_model, embeds, lens, guidance):
            self._train_serial += 1
            return (0.5 * latents.float()).to(torch.bfloat16)

        def backward(self, dv):
            assert torch.isfinite(dv).all() and float(dv.abs().sum()) > 0          # the scheduler-step adjoint produced d loss / d v
            for buf in self.engine.bufs.values():
                buf.fill_(float(rank + 1))                                        # this rank's micro-batch gradient

    eng = Eng()
    host = types.SimpleNamespace(engine=eng)
    host._live_weights = LiveWeights(eng, lambda: ddp)
    host._live_weights.sync()
    assert AG.unsupported_reason(host) is None
    g = torch.Generator().manual_seed(5 + rank)                                   # different data per rank
    B = 2
    x, x1 = torch.randn(B, 16, 64, generator=g).bfloat16(), torch.randn(B, 16, 64, generator=g).bfloat16()
    call = dict(latents=x, train_args=(x, torch.full((B,), 900.0), None, None, 1.0), sigma=torch.full((B,), 0.9), sigma_next=torch.full((B,), 0.75),
                eta=0.7, sigma_max=0.95, dynamics="Flow-SDE", next_latents=x1, compute_log_prob=True)
    trainable = [p for p in mod.parameters() if p.requires_grad]
    lp, npred, mean, std, dtt = AG.qwen_replay(host, Plan(eng), call)
    assert lp.requires_grad and lp.shape == (B,)
    lp.sum().backward()
    ok = all(p.grad is not None and torch.allclose(p.grad, torch.full_like(p.grad, 1.5)) for p in trainable)      # mean of the ranks' 1 and 2
    ok = ok and mod.get_submodule("img_in").weight.grad is None and eng.bufs == {}
    for p in trainable:
        p.grad = None
    with ddp.no_sync():                                                           # accumulation window: no reduction
        lp2 = AG.qwen_replay(host, Plan(eng), call)[0]
        lp2.sum().backward()
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p.grad, float(rank + 1))) for p in trainable)
    np.save(os.path.join(out_dir, f"qwen_ddp_{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


def test_qwen_replay_node_all_reduces_engine_gradients_under_ddp(tmp_path):
    """Round 4: the FLUX.1 / Qwen-Image replay (`mi355_flow.autograd.flux_replay` / `qwen_replay`: training-mode forward -> scheduler step ->
    step adjoint -> engine backward into registered buffers) under DistributedDataParallel on two ranks: the node arms the reducer itself,
    the engine-written gradients are bucket-all-reduced (mean over ranks), a frozen parameter outside the native scope stays out of both,
    and a `no_sync()` window accumulates locally.  Engine and plan are doubles of the host API; the scheduler step and its adjoint are the
    oracle's (`_plugin_fakes.oracle_sde_step[_bwd]`)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_qwen_replay_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(int(np.load(tmp_path / f"qwen_ddp_{r}.npy")[0]) == 1 for r in range(2))


def _fsdp2_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.distributed.device_mesh import init_device_mesh
    from torch.distributed.fsdp import fully_shard
    import _plugin_fakes as F
    from mi355_flow.binding import LiveWeights
    from oracle import qwen_ref as Q
    shapes = Q.state_dict_shapes(Q.tiny_config(num_layers=2, heads=1, joint_attention_dim=64))
    mod = F.build_module_tree(shapes, buffers=(), seed=3)                 # same seed on both ranks = the unsharded model
    want = {k: v.detach().clone() for k, v in mod.named_parameters()}
    mesh = init_device_mesh("cpu", (world,))
    for blk in mod.transformer_blocks.children():                          # per-block groups + the root, like accelerate's FSDP2 wrap
        fully_shard(blk, mesh=mesh)
    fully_shard(mod, mesh=mesh)

    class Recorder:
        def __init__(self):
            self.bound, self.finished = {}, 0

        def param_names(self):
            return list(shapes)

        def bind_tensor(self, name, t):
            assert not hasattr(t, "full_tensor"), "the engine must receive whole tensors, not shards"
            self.bound[name] = t.detach().float().clone()

        def finish_binding(self):
            self.finished += 1

    eng = Recorder()
    live = LiveWeights(eng, lambda: mod)
    n0 = live.sync()
    ok = n0 == len(shapes) and all(torch.equal(eng.bound[k], want[k]) for k in shapes)
    ok = ok and live.sync() == 0                                           # nothing changed -> nothing re-bound (no all-gathers)
    # one optimizer step on the SHARDED parameters: every rank updates its shard in place; the next sync must see it
    sharded = type(next(mod.parameters())).__name__ == "DTensor" and next(mod.parameters())._local_tensor.numel() < want[next(iter(shapes))].numel()
    for p in mod.parameters():
        p.grad = torch.ones_like(p)
    torch.optim.SGD(mod.parameters(), lr=0.5).step()
    n1 = live.sync()
    ok = ok and sharded and n1 == len(shapes) and all(torch.allclose(eng.bound[k], want[k] - 0.5) for k in shapes)
    np.save(os.path.join(out_dir, f"fsdp_{rank}.npy"), np.array([int(ok), n0, n1]))
    dist.barrier()
    dist.destroy_process_group()


def test_fsdp2_sharded_module_binds_whole_tensors_and_tracks_optimizer_steps(tmp_path):
    """SURVEY.md 8(f) N4, config E (Qwen-Image FSDP2, config/accelerate_configs/fsdp2.yaml): the trainable transformer is sharded with
    `fully_shard`; `LiveWeights.sync()` hands the engine WHOLE tensors (`DTensor.full_tensor()`, a collective all ranks enter together),
    skips everything unchanged, and re-binds after an in-place optimizer step on the shards (keyed on the DTensor's own version
    counter: the local shard's does not move)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_fsdp2_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok, n0, n1 = np.load(tmp_path / f"fsdp_{r}.npy")
        assert ok == 1 and n0 == n1 > 0, (r, ok, n0, n1)


def _trainer_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from torch.nn.parallel import DistributedDataParallel as DDP
    from oracle import ref_package
    ref_package.install()
    import _plugin_fakes as F
    import mi355_flow.flow_factory_plugin as P
    import test_plugin_binding as T
    from flow_factory.advantage.advantage_processor import AdvantageProcessor
    from flow_factory.trainers.grpo import GRPOTrainer
    P.Engine, P.VAEDecoder = F.FakeEngine, F.FakeVAEDecoder
    P.VAEConfig = types.SimpleNamespace(from_hf=lambda c: c)
    M_total, K = 4, 2
    M = M_total // world
    mine = T._prompt_batches(M_total, K, cfg_pair=True)[rank * M:(rank + 1) * M]       # GroupContiguousSampler: rank r owns whole groups

    def tweak(cfg):
        T._small(cfg.training_args, kl_beta=0.05, clip_range=(-1e-4, 1e-4), adv_clip_range=(-5.0, 5.0))
    acc = F.DistTrainerAccelerator()
    tr, ad, tr_mod, logged = T._real_trainer(P, GRPOTrainer, T.YAML_FULL, tweak, mine, K, accelerator=acc)
    # what BaseTrainer._initialization does with accelerator.prepare (trainers/abc.py:255-263): the trainable component becomes the DDP module
    ad.set_component("transformer", DDP(tr_mod))
    trainable = ad.get_trainable_parameters()
    before = torch.cat([p_.detach().reshape(-1) for p_ in trainable]).clone()
    torch.manual_seed(100 + rank)                     # device-specific seeding (trainers/loader.py:70): the ranks draw different noise
    samples = tr.sample()
    ok = len(samples) == M * K and len({s.unique_id for s in samples}) == M
    # ---- advantages: 2-rank result of the reference's processor == its single-process result on the union of the samples
    # K = 2: group-normalised advantages are (-1, +1) or (+1, -1).  One group has the same pattern on both ranks, the other MIRRORED ones:
    # the engine double's d(log-prob)/dW depends (almost only) on the sample's position in the batch, so the local gradients of the mirrored
    # micro-batch are non-zero and opposite -- their DDP average all but vanishes.  A collapsed all-reduced gradient norm there (next to a
    # full-size one for the other micro-batch) is the evidence that the reducer really averaged across ranks.
    all_r = torch.tensor([0.1, 0.9, 0.4, 0.2, 0.3, 0.7, 0.5, 0.8])
    adv = tr.compute_advantages(samples, {"r": all_r[rank * M * K:(rank + 1) * M * K]}, store_to_samples=True)
    ids = acc.gather(torch.tensor([s.unique_id for s in samples], dtype=torch.int64))
    union = [types.SimpleNamespace(unique_id=int(i), extra_kwargs={}) for i in ids]
    solo = AdvantageProcessor(accelerator=F.TrainerAccelerator(), reward_weights={"r": 1.0}, group_size=K, global_std=True,
                              sampler_type="group_contiguous", verbose=False)
    want = solo.compute_advantages(samples=union, rewards={"r": all_r}, store_to_samples=False,
                                   aggregation_func=tr.training_args.advantage_aggregation)
    ok_adv = bool(torch.allclose(adv.float().cpu(), torch.as_tensor(want).float()[rank * M * K:(rank + 1) * M * K], rtol=1e-5, atol=1e-6))
    # ---- optimize(): DDP averages the gradients the engine's autograd node returns; every rank takes the same steps
    n0 = len(ad.engine.calls)
    if os.environ.get("MI355_TEST_DEBUG"):
        _step = tr.optimizer.step
        def step(*a, **k):
            gs = [p_.grad for g_ in tr.optimizer.param_groups for p_ in g_["params"]]
            print(f"[rank {rank}] optimizer.step: {sum(g is not None for g in gs)}/{len(gs)} grads, max |g| = "
                  f"{max((float(g.abs().max()) for g in gs if g is not None), default=-1):.3e}; same objects as trainable: "
                  f"{all(a is b for a, b in zip([p_ for g_ in tr.optimizer.param_groups for p_ in g_['params']], trainable))}", flush=True)
            return _step(*a, **k)
        tr.optimizer.step = step
    tr.optimize(samples)
    kinds = [c[0] for c in ad.engine.calls[n0:]]
    first = logged[0][1]
    ok_ratio = first["train/ratio_min"] == 1.0 and first["train/ratio_max"] == 1.0
    after = torch.cat([p_.detach().reshape(-1) for p_ in trainable])
    both = acc.gather(after.reshape(1, -1))
    ok_same = bool(torch.equal(both[0], both[1])) and not torch.equal(after, before)
    if os.environ.get("MI355_TEST_DEBUG"):
        print(f"[rank {rank}] adv = {adv.tolist()} logged[0] = { {k: float(v) for k, v in logged[0][1].items()} }", flush=True)
        print(f"[rank {rank}] max |w0 - w1| = {float((both[0] - both[1]).abs().max()):.3e}, moved {float((after - before).abs().max()):.3e}, "
              f"ddp = {type(ad.transformer).__name__}, kinds = {kinds}", flush=True)
    ok_calls = kinds.count("denoise_step_train") == kinds.count("denoise_step_backward") >= M
    gn = sorted(float(d["train/grad_norm"]) for _, d in logged[:2])       # (optimize() shuffles with a seeded generator: same order on both ranks)
    ok_reduced = gn[1] > 0.0 and gn[0] < 0.05 * gn[1]
    # the next rollout of BOTH ranks runs on the same updated policy (no explicit re-bind)
    tr.epoch = 1
    tr.sample()
    fp = torch.tensor([[c for c in ad.engine.calls if c[0] in ("rollout", "denoise_step")][-1][1]["weights"]], dtype=torch.float64)
    fps = acc.gather(fp)
    ok_fp = bool(fps[0] == fps[1])
    np.save(os.path.join(out_dir, f"trainer_{rank}.npy"), np.array([int(ok), int(ok_adv), int(ok_ratio), int(ok_same), int(ok_calls), int(ok_fp),
                                                                    int(F.FakeTransformer.calls == 0), int(ok_reduced)]))
    dist.barrier()
    dist.destroy_process_group()


def test_reference_grpo_trainer_on_two_ranks_through_the_plugin(tmp_path):
    """SURVEY.md 8(e) at the trainer level: the reference's own `GRPOTrainer` (sample -> compute_advantages -> optimize), one process per
    rank on gloo, each rank rolling out ITS prompt groups through the plugin (no data-path collective), the reference's AdvantageProcessor
    gathering rewards across ranks (== its single-process result on the union), and the policy update all-reduced by DDP around the
    engine's autograd node: first ratio exactly 1 on every rank, bit-identical weights on both ranks after the epoch, the next rollout on
    the same new policy.  (Engine = the differentiable double; on the GPUs the same code runs with backend `nccl` = RCCL.)"""
    import socket
    import pytest
    from oracle import ref_package
    if not ref_package.available():
        pytest.skip("needs /root/reference (build container only)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_trainer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        flags = np.load(tmp_path / f"trainer_{r}.npy").tolist()
        assert flags == [1] * 8, (r, dict(zip(["samples", "advantages", "ratio", "same_weights", "calls", "next_policy", "no_torch_forward",
                                               "gradients_all_reduced"], flags)))


def _fsdp2_grad_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from torch.distributed.device_mesh import init_device_mesh
    from torch.distributed.fsdp import fully_shard
    import _plugin_fakes as F
    from mi355_flow import autograd as AG
    from mi355_flow.binding import LiveWeights
    from mi355_flow.engine import TransformerConfig
    from mi355_flow.weights import expected_shapes
    tcfg = TransformerConfig(num_layers=2, num_heads=1, joint_attention_dim=64, pooled_projection_dim=64, pos_embed_max_size=8, dual_layers=(0,))
    mod = F.build_module_tree(expected_shapes(tcfg), seed=3)                # same seed on both ranks = the unsharded model
    for n, p_ in mod.named_parameters():                                    # the reference's default target: the attention projections
        p_.requires_grad_(".attn.to_" in n or ".attn.add_" in n)
    mesh = init_device_mesh("cpu", (world,))
    for blk in mod.transformer_blocks.children():
        fully_shard(blk, mesh=mesh)
    fully_shard(mod, mesh=mesh)
    eng = F.DiffFakeEngine(tcfg)
    host = types.SimpleNamespace(engine=eng, _live_weights=LiveWeights(eng, lambda: mod))
    host._sync_weights = host._live_weights.sync
    assert AG.unsupported_reason(host) is None
    plan = eng.plan(2, 1, 8, 8, 5, 4)
    g = torch.Generator().manual_seed(10 + rank)                             # every rank has its own micro-batch
    lat = torch.randn(2, 16, 8, 8, generator=g).half()
    call = dict(latents=lat, timestep=torch.tensor([500.0, 500.0]), enc_a=torch.zeros(2, 5, 64), pooled_a=torch.zeros(2, 64), enc_b=None,
                pooled_b=None, guidance=1.0, sigma=0.5, sigma_next=0.4, eta=0.7, sigma_max=0.98, dynamics="Flow-SDE",
                next_latents=lat, compute_log_prob=True)
    lp, npred, mean, std, dtt = AG.denoise_replay(host, plan, call)
    wgt = torch.tensor([1.0, -3.0]) * (rank + 1)
    (wgt * lp).sum().backward()
    # what the engine double returns on THIS rank for every element of a signal tensor (see DiffFakePlan.denoise_step_backward)
    sig = eng.signal_names()
    m = lat.float().reshape(2, -1).mean(1)
    local_up = F.DiffFakePlan.C_W * float((wgt * (1.0 + m)).sum())
    ups = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(ups, torch.tensor([local_up], dtype=torch.float64))
    mean_up = float(sum(u.item() for u in ups) / world)                     # FSDP2 averages gradients over the data-parallel mesh
    ok, n_sig, n_sharded = True, 0, 0
    for name, p_ in mod.named_parameters():
        if not p_.requires_grad:
            ok = ok and p_.grad is None
            continue
        gshard = p_.grad
        ok = ok and gshard is not None and type(gshard).__name__ == "DTensor" and gshard.placements == p_.placements
        if not ok:
            break
        loc = gshard.to_local()
        n_sharded += int(loc.numel() < p_.numel())
        if name in sig:
            want = mean_up / (p_.numel() * len(sig))
            ok = ok and bool(torch.allclose(loc.double(), torch.full_like(loc, want, dtype=torch.float64), rtol=1e-5, atol=1e-12))
            n_sig += 1
        else:
            ok = ok and float(loc.abs().max()) == 0.0
    # an optimizer step on the shards is seen by the next sync (and both ranks hold the same model again)
    before = eng.weight_signal()
    torch.optim.SGD([p_ for p_ in mod.parameters() if p_.requires_grad], lr=1e4).step()
    host._live_weights.sync()
    after = eng.weight_signal()
    sigs = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sigs, torch.tensor([after], dtype=torch.float64))
    ok = ok and after != before and float(sigs[0]) == float(sigs[1])
    np.save(os.path.join(out_dir, f"fsdpgrad_{rank}.npy"), np.array([int(ok), n_sig, n_sharded]))
    dist.barrier()
    dist.destroy_process_group()


def test_fsdp2_sharded_parameters_receive_averaged_gradients_from_the_engine_autograd_node(tmp_path):
    """The reference's full fine-tuning example for SD3.5 runs under FSDP2 (examples/grpo/full/sd3_5/default.yaml): the trainable parameters
    are sharded DTensors, and the engine's autograd node -- which bypasses `module.forward`, so FSDP2's own reduce-scatter hooks never
    fire -- hands back WHOLE gradients of the local micro-batch.  `mi355_flow.autograd` bridges them: no all-gather in the forward, and in
    the backward the whole gradient is declared a partial value and redistributed to the parameter's placements (reduce-scatter with
    averaging).  Checked on 2 ranks with different micro-batches: every trainable shard's `.grad` is a DTensor with the parameter's
    placements holding the MEAN of the ranks' gradients; frozen parameters get none; an optimizer step on the shards reaches the engine."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_fsdp2_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok, n_sig, n_sharded = np.load(tmp_path / f"fsdpgrad_{r}.npy").tolist()
        assert ok == 1 and n_sig >= 4 and n_sharded > 0, (r, ok, n_sig, n_sharded)


def _fsdp2_lora_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from torch.distributed.device_mesh import init_device_mesh
    from torch.distributed.fsdp import fully_shard
    import _plugin_fakes as F
    from mi355_flow import autograd as AG
    from mi355_flow.binding import LiveWeights
    from mi355_flow.engine import TransformerConfig
    from mi355_flow.weights import expected_shapes
    tcfg = TransformerConfig(num_layers=2, num_heads=1, joint_attention_dim=64, pooled_projection_dim=64, pos_embed_max_size=8, dual_layers=(0,))

    def build():
        m = F.build_module_tree(expected_shapes(tcfg), seed=3)
        F.wrap_lora(m)
        for n, p_ in m.named_parameters():
            p_.requires_grad_("lora_" in n)
        return m
    mod, twin = build(), build()                                            # twin: the same model, unsharded, for the expected gradients
    mesh = init_device_mesh("cpu", (world,))
    for blk in mod.transformer_blocks.children():
        fully_shard(blk, mesh=mesh)
    fully_shard(mod, mesh=mesh)

    def run(m):
        eng = F.DiffFakeEngine(tcfg)
        host = types.SimpleNamespace(engine=eng, _live_weights=LiveWeights(eng, lambda: m))
        host._sync_weights = host._live_weights.sync
        assert AG.unsupported_reason(host) is None
        plan = eng.plan(2, 1, 8, 8, 5, 4)
        return host, plan

    def loss_of(host, plan, r):
        g = torch.Generator().manual_seed(10 + r)
        lat = torch.randn(2, 16, 8, 8, generator=g).half()
        call = dict(latents=lat, timestep=torch.tensor([500.0, 500.0]), enc_a=torch.zeros(2, 5, 64), pooled_a=torch.zeros(2, 64), enc_b=None,
                    pooled_b=None, guidance=1.0, sigma=0.5, sigma_next=0.4, eta=0.7, sigma_max=0.98, dynamics="Flow-SDE",
                    next_latents=lat, compute_log_prob=True)
        lp = AG.denoise_replay(host, plan, call)[0]
        return (torch.tensor([1.0, -3.0]) * (r + 1) * lp).sum()

    host, plan = run(mod)
    loss_of(host, plan, rank).backward()                                    # this rank's micro-batch on the SHARDED model
    thost, tplan = run(twin)
    for r in range(world):                                                  # every rank's micro-batch on the unsharded twin: the mean is expected
        (loss_of(thost, tplan, r) / world).backward()
    ok, n = True, 0
    want = dict(twin.named_parameters())
    for name, p_ in mod.named_parameters():
        if not p_.requires_grad:
            ok = ok and p_.grad is None
            continue
        gfull = p_.grad.full_tensor()
        ok = ok and type(p_.grad).__name__ == "DTensor" and bool(torch.allclose(gfull, want[name].grad, rtol=1e-5, atol=1e-9))
        n += int(float(want[name].grad.abs().max()) > 0)
    np.save(os.path.join(out_dir, f"fsdplora_{rank}.npy"), np.array([int(ok), n]))
    dist.barrier()
    dist.destroy_process_group()


def test_fsdp2_sharded_lora_factors_receive_averaged_gradients(tmp_path):
    """LoRA under FSDP2: the A / B factors are sharded DTensors whose VALUES enter the merged weight, so they are gathered with autograd
    (`full_tensor(grad_placements=[Partial("avg")])`): on 2 ranks with different micro-batches every factor's gradient equals the mean of
    the per-rank gradients computed on an unsharded twin of the model; the frozen base weights get none."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_fsdp2_lora_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok, n = np.load(tmp_path / f"fsdplora_{r}.npy").tolist()
        assert ok == 1 and n >= 4, (r, ok, n)
