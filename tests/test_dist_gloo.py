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
