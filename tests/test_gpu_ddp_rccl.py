"""RCCL under the engine's autograd node, on the one GPU a gpurun box has (reference trainers/grpo.py:326-330: the loss
goes through `accelerator.backward`, the trainable component is DDP-wrapped by `accelerator.prepare`, trainers/loader.py:33 /
trainers/abc.py).  A world-size-1 `nccl` process group is still RCCL: communicator init, the reducer's bucket views, the all-reduce kernels
launched on RCCL's stream behind the autograd engine's, the copy back into `.grad` -- everything except a second peer.  One `optimize()`
micro-step, written the way the reference trainer does it:

  DDP(module) -> adapter bound to the WRAPPED module -> rollout -> forward() WITH autograd (mi355_flow.autograd arms the reducer around
  `_DenoiseReplayFn`) -> PPO-clip loss -> backward (engine gradients, reduced bucket by bucket) -> clip_grad_norm -> AdamW -> the next
  forward sees the re-bound weights.

Checked against the same step WITHOUT DDP on a twin module (same values): first ratio exactly 1 on both, gradients bit-identical (an
average over one rank), `no_sync()` accumulates without reducing, the reducer really ran (its bucket hook fires once per bucket), and the
post-step replay log-prob moves identically.  The N > 1 behaviour of the same code is covered by the 2-rank gloo tests (tests/test_dist_gloo.py)."""
import os

import pytest
import torch

import _plugin_fakes as PF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def _make(seed=4):
    from mi355_flow.engine import TransformerConfig
    from mi355_flow.weights import expected_shapes
    cfg = TransformerConfig(num_layers=3, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24, dual_layers=(0, 1))
    mod = PF.build_module_tree(expected_shapes(cfg), seed=seed, std=0.08).cuda()
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(p.bfloat16().float())
    for n, p in mod.named_parameters():
        p.requires_grad_(any(k in n for k in (".to_q.", ".to_k.", ".to_v.", ".to_out.0.")))
    return cfg, mod


def test_one_optimize_microstep_under_ddp_on_rccl(rccl_group):
    from torch.nn.parallel import DistributedDataParallel as DDP
    from mi355_flow.adapter import SD3_5NativeAdapter
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    dist = rccl_group
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    cfg, mod_a = _make()
    _, mod_b = _make()
    # small buckets -> several all-reduces per backward (the default 25 MiB would take this tiny model in one)
    ddp = DDP(mod_a, device_ids=[0], bucket_cap_mb=0.05, broadcast_buffers=False)
    fired = []

    def hook(state, bucket):                      # torch's own allreduce_hook plus a counter (world size 1: the average is the sum)
        fired.append(bucket.buffer().numel())
        return torch.distributed.all_reduce(bucket.buffer(), async_op=True).get_future().then(lambda f: f.value()[0])
    ddp.register_comm_hook(None, hook)
    mk = lambda: FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[1, 2, 3], num_sde_steps=1, seed=42, shift=3.0)
    ad_a = SD3_5NativeAdapter(ddp, cfg, mk(), latent_storage_dtype="fp16")          # bound to the WRAPPED module, like the plugin
    ad_b = SD3_5NativeAdapter(mod_b, cfg, mk(), latent_storage_dtype="fp16")
    B, h, w, Nt, N = 4, 16, 16, 13, 4
    g = torch.Generator().manual_seed(3)
    pe, pp = torch.randn(B, Nt, 128, generator=g).bfloat16().cuda(), torch.randn(B, 128, generator=g).bfloat16().cuda()
    adv = torch.tensor([1.0, -0.5, 0.25, -0.75]).cuda()

    import contextlib

    def microstep(ad, sync_ctx=contextlib.nullcontext):
        ad.rollout()
        ad.scheduler.set_timesteps(N)
        x = torch.randn(B, 16, h, w, generator=torch.Generator().manual_seed(9)).half().cuda()
        t, t_next = torch.full((B,), 900.0), torch.full((B,), 750.0)
        torch.cuda.manual_seed(77)
        with torch.no_grad():
            o0 = ad.forward(t=t, t_next=t_next, latents=x, prompt_embeds=pe, pooled_prompt_embeds=pp, guidance_scale=1.0, noise_level=0.7,
                            return_kwargs=["next_latents", "log_prob"])
        ad.train()
        kw = dict(t=t, t_next=t_next, latents=x, next_latents=o0.next_latents.half(), prompt_embeds=pe, pooled_prompt_embeds=pp,
                  guidance_scale=1.0, noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob", "dt"])
        with sync_ctx():                          # accelerator.accumulate(...) wraps forward AND backward (trainers/grpo.py:236)
            out = ad.forward(**kw)
            ratio = torch.exp(out.log_prob - o0.log_prob)
            loss = torch.mean(torch.maximum(-adv * ratio, -adv * torch.clamp(ratio, 1 - 1e-4, 1 + 1e-4)))
            loss.backward()
        return ratio.detach(), kw

    params_a = [p for p in mod_a.parameters() if p.requires_grad]
    params_b = [p for p in mod_b.parameters() if p.requires_grad]
    opt_a, opt_b = torch.optim.AdamW(params_a, lr=1e-3), torch.optim.AdamW(params_b, lr=1e-3)

    # (1) accumulation window: no_sync() -> the reducer must NOT fire, gradients still land in .grad
    ra, _ = microstep(ad_a, sync_ctx=ddp.no_sync)
    assert fired == [], "no_sync() window must not all-reduce"
    assert all(p.grad is not None for p in params_a)
    rb, _ = microstep(ad_b)
    assert torch.equal(ra, torch.ones_like(ra)) and torch.equal(rb, torch.ones_like(rb))          # first ratio exactly 1, with and without DDP
    for pa, pb in zip(params_a, params_b):
        assert torch.equal(pa.grad, pb.grad)
    # (2) the synchronising micro-step: second half of the accumulation, reduced bucket by bucket on RCCL
    ra, kw_a = microstep(ad_a)
    rb, kw_b = microstep(ad_b)
    torch.cuda.synchronize()
    # (torch builds ONE bucket for the first reduction and re-buckets by `bucket_cap_mb` in gradient-ready order afterwards: part 4)
    assert len(fired) >= 1 and sum(fired) == sum(p.numel() for p in params_a), fired
    n_nonzero = 0
    for pa, pb in zip(params_a, params_b):
        assert torch.isfinite(pa.grad).all()
        assert torch.equal(pa.grad, pb.grad)              # accumulated 2 micro-steps; average over ONE rank == identity, bit for bit
        n_nonzero += int(float(pa.grad.abs().max()) > 0)
    assert n_nonzero >= len(params_a) - 4                 # (the context-pre-only last block's add_q_proj has an exactly zero gradient)
    for ps, opt in ((params_a, opt_a), (params_b, opt_b)):
        torch.nn.utils.clip_grad_norm_(ps, 1.0)
        opt.step()
        opt.zero_grad()
    # (3) the next forward runs on the updated weights (live re-bind through the DDP wrapper), identically on both
    with torch.no_grad():
        la = ad_a.forward(**kw_a).log_prob
        lb = ad_b.forward(**kw_b).log_prob
    assert torch.equal(la, lb)
    # (4) the next synchronising micro-step runs on the REBUILT buckets (bucket_cap_mb = 0.05: many all-reduces, each launched on RCCL's
    #     stream while the engine's backward is still producing the later gradients), on the updated weights; still bit-identical to the twin
    n_before = len(fired)
    ra, _ = microstep(ad_a)
    rb, _ = microstep(ad_b)
    torch.cuda.synchronize()
    assert len(fired) - n_before >= 2, fired
    assert sum(fired[n_before:]) == sum(p.numel() for p in params_a)
    assert torch.equal(ra, rb)                  # (each micro-step samples its own transition on the CURRENT weights: ratio 1 on both)
    for pa, pb in zip(params_a, params_b):
        assert torch.equal(pa.grad, pb.grad)
    print(f"RCCL world-size-1 DDP optimize() micro-step: {len(fired)} buckets / {sum(fired)} gradient elements all-reduced on backend "
          f"{dist.get_backend()}; first ratio == 1; gradients bit-identical to the unwrapped twin; post-step log-prob {la.tolist()}")
    ad_a.engine.close()
    ad_b.engine.close()
