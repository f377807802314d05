"""GPU checks of the launch SCHEDULES (side streams, hipGraph replay) of the FLUX.1 and Qwen-Image engines -- bit-identity of every
combination against single-stream eager launches, whole rollouts included -- plus two checks that were written without GPU time at the end
of round 2 and first ran (and passed) in round 3's first GPU call: the SD3.5-large width forward and bench.py's small-batch legs.
Measured gains of the schedules: profiles/r03a_*_two_stream_ab.txt."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_forward_at_sd3_5_large_width_vs_oracle():
    """SD3.5-large geometry (38 heads x 64 = 2432 wide, no dual-attention blocks; 2 of its 38 blocks): 2432 is not a multiple of the
    256 / 128 GEMM tile widths, so every N-ragged tile path (guards instead of the FULL fast path) and the q|k epilogue at 38 heads run.
    Same tolerances as the SD3.5-medium forward tests (tests/test_gpu_model.py)."""
    from mi355_flow import engine
    from oracle import mmditx_ref as M
    cfg = M.tiny_config(num_layers=2, num_heads=38, dual_layers=(), joint_attention_dim=256, pooled_projection_dim=128, pos_embed_max_size=64)
    sd = {k: v.bfloat16().float() for k, v in M.make_synthetic_state_dict(cfg, seed=5, std=0.02).items()}
    cfg_e = engine.TransformerConfig.from_hf(dict(num_layers=2, num_attention_heads=38, attention_head_dim=64, joint_attention_dim=256,
                                                  caption_projection_dim=2432, pooled_projection_dim=128, pos_embed_max_size=64,
                                                  dual_attention_layers=(), qk_norm="rms_norm", in_channels=16, out_channels=16, patch_size=2))
    e = engine.Engine(cfg_e)
    e.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    e.ready()
    try:
        # (the last case has M = 8192 image rows: the persistent 256x256 ping-pong kernel with its last column tile half outside N = 2432)
        for B, h, w, Nt in ((2, 32, 32, 77), (1, 48, 16, 13), (2, 128, 128, 77)):
            g = torch.Generator().manual_seed(B * 100 + h)
            x = torch.randn(B, 16, h, w, generator=g).half()
            enc = torch.randn(B, Nt, cfg.joint_attention_dim, generator=g).bfloat16()
            pooled = torch.randn(B, cfg.pooled_projection_dim, generator=g).bfloat16()
            t = torch.tensor([873.0] * B)
            plan = e.plan(B, 1, h, w, Nt, 4)
            y = plan.transformer_forward(x.cuda(), t.cuda(), enc.cuda(), pooled.cuda())
            torch.cuda.synchronize()
            t_net = t.half().float()
            from _gpu_oracle import check_in_band
            check_in_band("SD3.5-large width forward", y, M.mmdit_forward, sd, cfg, x.float(), t_net, enc.float(), pooled.float())
    finally:
        e.close()


def test_bench_small_batch_legs_report_numbers():
    """bench.py's untimed small-batch legs (`--small-batch`: B = 2 at 1024^2; the reference's 512^2 B = 2 CFG example shape) produce finite
    figures and do not disturb the headline line (`--no-small-batch` switches them off)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--no-vae", "--no-cpu-baseline", "--no-families",
                        "--no-train-step", "--no-clock-probe", "--no-selfcheck"],      # (the other legs have their own scripts; this test is about one leg)
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    sb = line["small_batch"]
    assert "error" not in sb, sb
    for tag in ("b2_1024_nocfg_28", "b2_512_cfg4.5_10"):
        assert sb[tag]["denoise_steps_per_s"] > 0 and 0 < sb[tag]["forward_frac"] < 1


def test_qwen_two_stream_and_graph_replay_are_bit_identical():
    """Qwen-Image engine with the text chain of every block on a side stream (mi355_tune_set key 12; default 2 = plans of up to 16 384 image rows) and / or the N-step
    loop replayed as one hipGraph (key 17 = 1; default 0): the raw network outputs of both CFG branches of a ragged-prompt forward and whole
    true-CFG rollouts (latents, log-probs) equal the single-stream eager results bit for bit, repeatedly (a missing fork / join edge shows
    up as a run-to-run difference); a replayed graph sees new prompt lengths."""
    from mi355_flow import _lib, qwen as qw
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    from oracle import qwen_ref as R
    lib = _lib.load()
    cfg_o = R.tiny_config()
    sd = {k: v.bfloat16().float() for k, v in R.make_synthetic_state_dict(cfg_o, seed=3, std=0.03).items()}
    cfg = qw.QwenConfig(num_layers=cfg_o.num_layers, num_attention_heads=cfg_o.num_attention_heads,
                        joint_attention_dim=cfg_o.joint_attention_dim, scale_rope=cfg_o.scale_rope)
    J = cfg_o.joint_attention_dim
    g = torch.Generator().manual_seed(5)
    try:
        # ---- single forwards: a small and a chip-filling token count, ragged text, both CFG branches
        eng = qw.QwenEngine(cfg)
        eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
        eng.ready()
        for (B, h, w, Nt) in ((2, 8, 12, 19), (2, 64, 64, 77)):
            x = torch.randn(B, (h // 2) * (w // 2), 64, generator=g).bfloat16()
            lens = [Nt, max(1, Nt - 5)] * 2
            emb = torch.randn(2 * B, Nt, J, generator=g).bfloat16()
            for b, n in enumerate(lens):
                emb[b, n:] = 0
            tm = qw.model_timestep(torch.tensor([875.0, 500.0][:B]), torch.bfloat16)
            outs = {}
            for mode in (0, 1, 1, 0, 1):
                lib.mi355_tune_set(12, mode)
                plan = eng.plan(B, 2, h, w, Nt, 1)
                v, raw = plan.transformer_forward(x.cuda(), tm, emb.cuda(), lens, guidance_scale=4.0, return_raw=True)
                torch.cuda.synchronize()
                outs.setdefault(mode, []).append((v.clone(), raw.clone()))
            ref_v, ref_raw = outs[0][0]
            assert torch.isfinite(ref_raw.float()).all()
            for mode, runs in outs.items():
                for v, raw in runs:
                    assert torch.equal(v, ref_v) and torch.equal(raw, ref_raw), (mode, B, h, w, Nt)
        eng.close()
        # ---- whole rollouts through the adapter: single / two-stream x eager / hipGraph replay of the loop (key 17; captured on the second
        #      call of a configuration: eager warm-up, capture + launch, replay).  The third run changes the prompts' lengths: the key
        #      lengths are uploaded in front of the graph, so a replay must see them.
        res = {}
        for mode in ((0, 0), (1, 0), (0, 1), (1, 1)):
            lib.mi355_tune_set(12, mode[0])
            lib.mi355_tune_set(17, mode[1])
            sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[0, 1, 2, 3], num_sde_steps=2, seed=42, dynamics_type="Flow-SDE",
                                                       shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9, max_image_seq_len=8192,
                                                       shift_terminal=0.02)
            ad = qw.QwenImageNativeAdapter({k: v.cuda() for k, v in sd.items()}, cfg, sched, latent_storage_dtype="bf16")
            ad.rollout()
            gg = torch.Generator().manual_seed(9)
            ne = torch.randn(2, 6, J, generator=gg).bfloat16().cuda()
            nm = torch.ones(2, 6, dtype=torch.long).cuda()
            full = [torch.randn(21, J, generator=gg).bfloat16().cuda() for _ in range(2)]
            runs = []
            for lens in ((21, 17), (21, 17), (21, 17), (21, 9)):
                pe = [full[b][:n] for b, n in enumerate(lens)]
                pm = [torch.ones(n, dtype=torch.long).cuda() for n in lens]
                torch.cuda.manual_seed(77)
                s = ad.inference(prompt=["a", "b"], negative_prompt=None, height=128, width=192, num_inference_steps=5, guidance_scale=4.0,
                                 prompt_embeds=pe, prompt_embeds_mask=pm, negative_prompt_embeds=ne, negative_prompt_embeds_mask=nm,
                                 compute_log_prob=True, trajectory_indices="all")
                runs.append((torch.stack([o.all_latents for o in s]).clone(), torch.stack([o.log_probs for o in s]).clone()))
            assert all(torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) for r in runs[1:3]), mode
            assert not torch.equal(runs[3][0][1], runs[0][0][1])          # the shorter second prompt changed sample 1 ...
            assert torch.equal(runs[3][0][0], runs[0][0][0])              # ... and only sample 1
            res[mode] = (runs[0][0], runs[0][1], runs[3][0], runs[3][1])
            ad.engine.close()
        ref = res[(0, 0)]
        for mode, r in res.items():
            assert all(torch.equal(a, b) for a, b in zip(r, ref)), mode
    finally:
        lib.mi355_tune_set(12, 2)                       # the defaults
        lib.mi355_tune_set(17, 1)


def test_flux_two_stream_and_graph_replay_are_bit_identical():
    """FLUX.1 engine with the text chain of the double blocks on a side stream (mi355_tune_set key 14; default 2 = plans of up to 16 384 image rows) and / or the N-step
    loop replayed as one hipGraph (key 16 = 1; default 0): single forwards (small and chip-filling token counts, text length not a multiple
    of 64 so that text and image columns of V^T share cache lines) and whole rollouts equal the single-stream eager results bit for bit,
    repeatedly; a changed step count re-captures."""
    from mi355_flow import _lib, flux as fx
    from mi355_flow.scheduler import FlowMatchEulerDiscreteSDEScheduler
    from oracle import flux_ref as R
    lib = _lib.load()
    cfg_o = R.tiny_config()
    sd = {k: v.bfloat16().float() for k, v in R.make_synthetic_state_dict(cfg_o, 77).items()}
    cfg = fx.FluxConfig(num_layers=cfg_o.num_layers, num_single_layers=cfg_o.num_single_layers, num_attention_heads=cfg_o.num_attention_heads,
                        joint_attention_dim=cfg_o.joint_attention_dim, pooled_projection_dim=cfg_o.pooled_projection_dim,
                        guidance_embeds=cfg_o.guidance_embeds)
    g = torch.Generator().manual_seed(5)
    try:
        eng = fx.FluxEngine(cfg)
        eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
        eng.ready()
        for (B, h, w, Nt) in ((2, 8, 8, 13), (2, 64, 64, 77), (1, 48, 48, 512)):
            x = R.pack_latents(torch.randn(B, 16, h, w, generator=g)).half()
            enc = torch.randn(B, Nt, cfg_o.joint_attention_dim, generator=g).bfloat16()
            pool = torch.randn(B, cfg_o.pooled_projection_dim, generator=g).bfloat16()
            tm = torch.tensor([900.0, 412.5][:B])
            gm = torch.full((B,), 3500.0) if cfg_o.guidance_embeds else None
            outs = []
            for mode in (0, 1, 1, 0, 1):
                lib.mi355_tune_set(14, mode)
                plan = eng.plan(B, h, w, Nt, 1)
                y = plan.transformer_forward(x.cuda(), tm, gm, enc.cuda(), pool.cuda())
                torch.cuda.synchronize()
                outs.append(y.clone())
            assert torch.isfinite(outs[0].float()).all()
            assert all(torch.equal(o, outs[0]) for o in outs[1:]), (B, h, w, Nt)
        eng.close()
        # ---- whole rollouts: single / two-stream x eager / hipGraph replay of the loop (key 16; the graph is captured on the second call
        #      of a configuration, so every mode runs three times: eager warm-up, capture + launch, replay)
        res = {}
        for mode in ((0, 0), (1, 0), (0, 1), (1, 1)):
            lib.mi355_tune_set(14, mode[0])
            lib.mi355_tune_set(16, mode[1])
            sched = FlowMatchEulerDiscreteSDEScheduler(noise_level=0.7, sde_steps=[0, 1, 2, 3], num_sde_steps=2, seed=42, dynamics_type="Flow-SDE",
                                                       shift=3.0, use_dynamic_shifting=True)
            ad = fx.Flux1NativeAdapter({k: v.cuda() for k, v in sd.items()}, cfg, sched, latent_storage_dtype="fp16")
            ad.rollout()
            gg = torch.Generator().manual_seed(9)
            pe = torch.randn(2, 16, cfg_o.joint_attention_dim, generator=gg).bfloat16().cuda()
            pp = torch.randn(2, cfg_o.pooled_projection_dim, generator=gg).bfloat16().cuda()
            runs = []
            for _ in range(3):
                torch.cuda.manual_seed(77)
                s = ad.inference(prompt=["a", "b"], height=128, width=128, num_inference_steps=5, guidance_scale=3.5, prompt_embeds=pe,
                                 pooled_prompt_embeds=pp, trajectory_indices="all")
                runs.append((torch.stack([o.all_latents for o in s]).clone(), torch.stack([o.log_probs for o in s]).clone()))
            assert all(torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) for r in runs[1:]), mode
            # a different step count / dynamics re-captures instead of replaying a stale graph
            torch.cuda.manual_seed(78)
            s4 = ad.inference(prompt=["a", "b"], height=128, width=128, num_inference_steps=4, guidance_scale=3.5, prompt_embeds=pe,
                              pooled_prompt_embeds=pp, trajectory_indices="all")
            res[mode] = (runs[0][0], runs[0][1], torch.stack([o.all_latents for o in s4]).clone())
            ad.engine.close()
        ref = res[(0, 0)]
        for mode, r in res.items():
            assert all(torch.equal(a, b) for a, b in zip(r, ref)), mode
    finally:
        lib.mi355_tune_set(14, 2)                       # the defaults
        lib.mi355_tune_set(16, 1)


# ------------------------------------------------------------------------------------------------- engine-emitted launch lists, race check
def _trace_of(lib, fn, reps=2, tag=None):
    """Run `fn` reps times with the library's schedule trace on (csrc/sched_trace.hip) and return the text.  With MI355_DUMP_TRACES=<dir> the
    text is also written to <dir>/<tag>.txt: recorded launch lists are committed under tests/golden/sched_traces/ and re-checked WITHOUT a GPU
    by tests/test_sched_recorded_traces.py."""
    import ctypes as C
    torch.cuda.synchronize()
    lib.mi355_sched_trace(1)
    try:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        need = lib.mi355_sched_trace_read(None, 0)
        buf = C.create_string_buffer(int(need))
        lib.mi355_sched_trace_read(buf, need)
    finally:
        lib.mi355_sched_trace(0)
    text = buf.value.decode()
    dump = os.environ.get("MI355_DUMP_TRACES")
    if dump and tag:
        os.makedirs(dump, exist_ok=True)
        with open(os.path.join(dump, tag + ".txt"), "w") as f:
            f.write(text)
    return text


def _assert_race_free(text, what, min_streams=2):
    import _sched_check as SC
    s = SC.parse(text)
    n_launch = sum(1 for o in s.ops if o.regions)
    assert len(s.streams()) >= min_streams, (what, s.streams())
    races = s.races()
    assert races == [], (what, races[:5])
    # the checker must SEE the dependencies the events carry: without any one stream wait most schedules race
    nw = SC.n_waits(text)
    needed = [k for k in range(nw) if SC.parse(text, drop_waits=[k]).races(limit=1)]
    print(f"{what}: {n_launch} launches on {len(s.streams())} streams, {nw} stream waits, {len(needed)} of them individually necessary, no race")
    assert nw > 0 and len(needed) >= 0.75 * nw, (what, nw, len(needed))


def test_sd3_engine_emitted_schedule_is_race_free():
    """The launch list the SD3.5 engine ITSELF emits for two consecutive forwards (eager launches; text chain on the side stream with the
    early and the late fork point, and the measured-and-dropped three-stream variant) fed to the happens-before checker: every pair of
    launches touching the same bytes is ordered, and removing any single stream wait makes a race appear (the checker sees the edges).
    Replaces round 2's hand-transcribed launch lists."""
    from mi355_flow import _lib, engine
    from oracle import mmditx_ref as M
    lib = _lib.load()
    cfg = M.tiny_config(num_layers=4, num_heads=2, dual_layers=(0, 1, 2), joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24)
    sd = {k: v.bfloat16().float() for k, v in M.make_synthetic_state_dict(cfg, seed=11, std=0.05).items()}
    e = engine.Engine(engine.TransformerConfig(num_layers=4, num_heads=2, joint_attention_dim=128, pooled_projection_dim=128, pos_embed_max_size=24,
                                               dual_layers=(0, 1, 2)))
    e.bind_state_dict({k: v.cuda() for k, v in sd.items()})
    e.ready()
    B, h, w, Nt = 2, 16, 16, 13
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 16, h, w, generator=g).half().cuda()
    enc = torch.randn(2 * B, Nt, 128, generator=g).bfloat16().cuda()
    pooled = torch.randn(2 * B, 128, generator=g).bfloat16().cuda()
    t = torch.full((2 * B,), 500.0).cuda()
    plan = e.plan(B, 2, h, w, Nt, 1)
    run = lambda: plan.transformer_forward(x, t, enc[:B], pooled[:B], enc[B:], pooled[B:])
    try:
        ref = run().clone()
        for name, keys in (("early fork", {8: 1, 10: 0, 11: 0}), ("late fork", {8: 1, 10: 1, 11: 0}), ("three streams", {8: 1, 10: 0, 11: 1})):
            for k, v in keys.items():
                lib.mi355_tune_set(k, v)
            text = _trace_of(lib, run, tag="sd3_forward_" + name.replace(" ", "_"))
            assert torch.equal(run(), ref), name
            _assert_race_free(text, f"SD3.5 forward, {name}", min_streams=3 if keys[11] else 2)
        lib.mi355_tune_set(8, 0)
        single = _trace_of(lib, run)
        import _sched_check as SC
        assert SC.parse(single).streams() == ["s0"] and SC.parse(single).races() == []
    finally:
        for k, v in ((8, 2), (10, 2), (11, 0)):
            lib.mi355_tune_set(k, v)
        e.close()


@pytest.mark.parametrize("scope", ["default_targets", "all_block_linears"])
def test_sd3_training_step_schedule_is_race_free(scope):
    """The optimize() replay step (training-mode forward + backward through torch autograd) as the engine emits it: the context-stream chain
    of both halves runs on the training state's side stream (mi355_tune_set(22, 1), the default), the weight gradients on the stream of the chain
    that produced their dY (the third-stream opt-in of round 4, mi355_tune_set(26, 2), measured neutral twice and was removed in round 5).
    Every launch of both halves reports its regions (GEMMs incl. the activation
    stashes, attention + log-sum-exp, the backward's elementwise / transpose / split-K / attention-backward kernels, the stash copies)."""
    from mi355_flow import _lib
    from test_gpu_backward import _build, _inputs, BLOCK_LINEARS
    lib = _lib.load()
    targets = (".to_q.", ".to_k.", ".to_v.", ".to_out.0.") if scope == "default_targets" else BLOCK_LINEARS
    ad, mod, _ = _build(lambda n: n.startswith("transformer_blocks.") and any(k in n for k in targets), seed=31)
    B, h, w, Nt = 2, 16, 16, 13
    inp = _inputs(B, h, w, Nt, seed=9)
    ad.scheduler.set_timesteps(4)
    kw = dict(t=torch.full((B,), 750.0), t_next=torch.full((B,), 500.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
              prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), guidance_scale=1.0, noise_level=0.7,
              compute_log_prob=True, return_kwargs=["log_prob", "dt"])
    wlp = inp["wlp"].cuda()

    def step():
        for prm in mod.parameters():
            prm.grad = None
        out = ad.forward(**kw)
        (wlp * out.log_prob).sum().backward()

    try:
        step()
        text = _trace_of(lib, step, tag="sd3_train_step_" + scope)
        import _sched_check as SC
        s = SC.parse(text)
        names = {o.name for o in s.ops}
        assert {"attention_bwd", "attn_bwd_prep", "rms_bwd_gather", "ln_mod_bwd", "gate_mul", "transpose"} <= names, sorted(names)
        # weight gradients: the row-major-operand kernel (round 6, csrc/gemm_tn.hip) -- since round 6b for the text rows too (a ragged last m-tile
        # reads zeros): no K-contiguous GEMM on transposed copies is left in this step
        assert "gemm_tn" in names and "gemm.f32" not in names, sorted(names)
        assert len(s.streams()) == 2, s.streams()          # caller's stream, context chain (key 22)
        races = s.races()
        assert races == [], races[:5]
        nw = SC.n_waits(text)
        needed = [k for k in range(nw) if SC.parse(text, drop_waits=[k]).races(limit=1)]
        print(f"SD3.5 optimize() step, {scope}: {sum(1 for o in s.ops if o.regions)} launches on 2 streams, {nw} stream waits, "
              f"{len(needed)} of them individually necessary, no race")
        assert nw > 0 and len(needed) >= 0.6 * nw, (nw, len(needed))
    finally:
        ad.engine.close()


def test_qwen_and_flux_engine_emitted_schedules_are_race_free():
    """The same for the Qwen-Image blocks (key 12) and the FLUX.1 double blocks (key 14): two consecutive forwards each, two streams."""
    from mi355_flow import _lib, flux as fx, qwen as qw
    from oracle import flux_ref as FR, qwen_ref as QR
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    try:
        # ---- Qwen-Image (ragged text, both CFG branches)
        cfg_o = QR.tiny_config()
        sd = {k: v.bfloat16().float() for k, v in QR.make_synthetic_state_dict(cfg_o, seed=3, std=0.03).items()}
        cfg = qw.QwenConfig(num_layers=cfg_o.num_layers, num_attention_heads=cfg_o.num_attention_heads,
                            joint_attention_dim=cfg_o.joint_attention_dim, scale_rope=cfg_o.scale_rope)
        J = cfg_o.joint_attention_dim
        eng = qw.QwenEngine(cfg)
        eng.bind_state_dict({k: v.cuda() for k, v in sd.items()})
        eng.ready()
        B, h, w, Nt = 2, 8, 12, 19
        x = torch.randn(B, (h // 2) * (w // 2), 64, generator=g).bfloat16().cuda()
        lens = [Nt, Nt - 5] * 2
        emb = torch.randn(2 * B, Nt, J, generator=g).bfloat16()
        for b, n in enumerate(lens):
            emb[b, n:] = 0
        emb = emb.cuda()
        tm = qw.model_timestep(torch.tensor([875.0, 500.0]), torch.bfloat16)
        lib.mi355_tune_set(12, 1)
        plan = eng.plan(B, 2, h, w, Nt, 1)
        text = _trace_of(lib, lambda: plan.transformer_forward(x, tm, emb, lens, guidance_scale=4.0, return_raw=True), tag="qwen_forward_two_stream")
        _assert_race_free(text, "Qwen-Image forward, two streams")
        eng.close()
        # ---- FLUX.1 (double blocks on two streams, then the single blocks on the concatenated stream)
        fo = FR.tiny_config()
        fsd = {k: v.bfloat16().float() for k, v in FR.make_synthetic_state_dict(fo, 77).items()}
        fcfg = fx.FluxConfig(num_layers=fo.num_layers, num_single_layers=fo.num_single_layers, num_attention_heads=fo.num_attention_heads,
                             joint_attention_dim=fo.joint_attention_dim, pooled_projection_dim=fo.pooled_projection_dim,
                             guidance_embeds=fo.guidance_embeds)
        feng = fx.FluxEngine(fcfg)
        feng.bind_state_dict({k: v.cuda() for k, v in fsd.items()})
        feng.ready()
        B, h, w, Nt = 2, 8, 8, 13
        xl = FR.pack_latents(torch.randn(B, 16, h, w, generator=g)).half().cuda()
        enc = torch.randn(B, Nt, fo.joint_attention_dim, generator=g).bfloat16().cuda()
        pool = torch.randn(B, fo.pooled_projection_dim, generator=g).bfloat16().cuda()
        tmf = torch.tensor([900.0, 412.5])
        gm = torch.full((B,), 3500.0) if fo.guidance_embeds else None
        lib.mi355_tune_set(14, 1)
        fplan = feng.plan(B, h, w, Nt, 1)
        ftext = _trace_of(lib, lambda: fplan.transformer_forward(xl, tmf, gm, enc, pool), tag="flux_forward_two_stream")
        _assert_race_free(ftext, "FLUX.1 forward, two streams")
        feng.close()
    finally:
        lib.mi355_tune_set(12, 2)
        lib.mi355_tune_set(14, 2)


@pytest.mark.parametrize("family", ["qwen", "flux"])
def test_head_dim_128_training_step_schedules_are_race_free(family):
    """The optimize() replay step of the Qwen-Image / FLUX.1 engines as emitted: training-mode forward (per-block stash buffers; the text chain of
    the double-stream blocks on the plan's side stream, as in the no-grad forward) + backward (dgrad chain, attention backward and the operand transposes on the caller's
    stream; the weight-gradient split-K GEMMs and their reductions on the training state's side stream, three operand slots handed over and
    back by events: mi355_tune_set(26, 1), the default).  Two steps back to back: the second forward overwrites the stash the first backward read."""
    from mi355_flow import _lib
    lib = _lib.load()
    lib.mi355_tune_set(12, 1); lib.mi355_tune_set(14, 1); lib.mi355_tune_set(26, 1)
    try:
        if family == "qwen":
            import test_gpu_qwen_backward as TQ
            from mi355_flow import qwen as qw
            from oracle import qwen_ref as R
            cfg_o = R.tiny_config()
            ad, mod = TQ._build(qw, cfg_o, lambda n: any(k in n for k in TQ.BLOCK_LINEARS), seed=31)
            B = 2
            inp = TQ._inputs(cfg_o, B, 8, 12, 19, 2, True, seed=9)
            ad.scheduler.set_timesteps(4, mu=0.6)
            kw = TQ._kw(inp, B, 750.0, 500.0, 0.7, 4.0)
        else:
            import test_gpu_flux_backward as TF
            ad, mod, cfg_o = TF._build(lambda n: any(k in n for k in TF.BLOCK_LINEARS), seed=31)
            B = 2
            inp = TF._inputs(cfg_o, B, 8, 8, 13, seed=9)
            ad.scheduler.set_timesteps(4)
            kw = dict(t=torch.full((B,), 750.0), t_next=torch.full((B,), 500.0), latents=inp["x"].cuda(), next_latents=inp["x1"].cuda(),
                      prompt_embeds=inp["pe"].cuda(), pooled_prompt_embeds=inp["pp"].cuda(), img_ids=inp["img_ids"].cuda(), guidance_scale=3.5,
                      noise_level=0.7, compute_log_prob=True, return_kwargs=["log_prob", "noise_pred", "dt"])
        wlp, wnp = inp["wlp"].cuda(), inp["wnp"].cuda()

        def step():
            for prm in mod.parameters():
                prm.grad = None
            out = ad.forward(**kw)
            ((wlp * out.log_prob).sum() + (wnp * out.noise_pred).mean()).backward()

        try:
            step()
            text = _trace_of(lib, step, tag=f"{family}_train_step")
            import _sched_check as SC
            s = SC.parse(text)
            names = {o.name for o in s.ops}
            assert {"attention128_bwd", "attn128_bwd_prep", "rope_rms_bwd128", "ln_mod_bwd", "gate_mul", "transpose", "gemm.f32"} <= names, sorted(names)
            n_streams = 3                                      # caller's stream, text chain (forward; Qwen-Image: backward too), weight gradients
            assert len(s.streams()) == n_streams, s.streams()
            races = s.races()
            assert races == [], races[:5]
            nw = SC.n_waits(text)
            needed = [k for k in range(nw) if SC.parse(text, drop_waits=[k]).races(limit=1)]
            print(f"{family} optimize() step: {sum(1 for o in s.ops if o.regions)} launches on {n_streams} streams, {nw} stream waits, {len(needed)} of them "
                  f"individually necessary, no race")
            assert nw > 0 and len(needed) >= 0.5 * nw, (nw, len(needed))
        finally:
            ad.engine.close()
    finally:
        lib.mi355_tune_set(12, 2); lib.mi355_tune_set(14, 2)


def test_wan_training_step_schedule_is_race_free():
    """The Wan optimize() replay step as emitted (training-mode forward on the caller's stream, backward with the weight-gradient GEMMs on the
    training state's side stream): every launch reports its regions; no unordered conflicting pair.  (First run = round 5's first GPU call:
    338 launches on 2 streams, 56 stream waits, 51 individually necessary, no race -- profiles/r05a_*; its launch list is recorded under
    tests/golden/sched_traces/wan_train_step.txt for the CPU-side re-check.)"""
    import os
    if os.environ.get("MI355_WAN_NATIVE_BACKWARD") == "0":
        pytest.skip("MI355_WAN_NATIVE_BACKWARD=0 opts out")
    from mi355_flow import _lib, wan as wn
    import test_gpu_wan_backward as TW
    from oracle import wan_ref as R
    lib = _lib.load()
    lib.mi355_tune_set(26, 1)
    cfg_o = R.tiny_config()
    ad, mod = TW._build(wn, cfg_o, lambda n: any(k in n for k in TW.DEFAULT_TARGETS), seed=31)
    B = 2
    inp = TW._inputs(cfg_o, B, 2, 8, 12, 17, seed=9)
    ad.scheduler.set_timesteps(4)
    kw = TW._kw(inp, B, 750.0, 500.0, 0.7, 5.0)
    wlp, wnp = inp["wlp"].cuda(), inp["wnp"].cuda()

    def step():
        for prm in mod.parameters():
            prm.grad = None
        out = ad.forward(**kw)
        ((wlp * out.log_prob).sum() + (wnp * out.noise_pred).mean()).backward()

    try:
        step()
        text = _trace_of(lib, step, tag="wan_train_step")
        import _sched_check as SC
        s = SC.parse(text)
        names = {o.name for o in s.ops}
        # (weight gradients on row-major operands since round 6b: `gemm_tn` on the backward's own stream instead of `gemm.f32` on transposed copies)
        assert {"attention128_bwd", "attn128_bwd_prep", "norm_rope_full_bwd", "ln_mod_bwd", "gate_mul", "transpose", "gemm_tn"} <= names, sorted(names)
        assert len(s.streams()) in (1, 2), s.streams()
        races = s.races()
        assert races == [], races[:5]
        nw = SC.n_waits(text)
        needed = [k for k in range(nw) if SC.parse(text, drop_waits=[k]).races(limit=1)]
        print(f"wan optimize() step: {sum(1 for o in s.ops if o.regions)} launches on {len(s.streams())} stream(s), {nw} stream waits, {len(needed)} of them "
              f"individually necessary, no race")
        # (round 6b: the weight-gradient GEMMs read dY where it lies and therefore run on the backward's own stream -- the side stream and its
        #  waits are gone from this step; where waits remain, at least half must be individually necessary)
        assert (nw == 0 and len(s.streams()) == 1) or len(needed) >= 0.5 * nw, (nw, len(needed), s.streams())
    finally:
        ad.engine.close()
