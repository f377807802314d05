"""TEST INFRASTRUCTURE (not product code).  Generates tests/golden/config_b_oracle.npz: the ORACLE side of tests/test_gpu_fullsize.py's config-B
checks (BASELINE.json configs[1]'s geometry end to end: SD3.5-medium, 1024^2, B = 1, N = 4 Flow-SDE steps) -- the fp32 oracle rollout, the
bf16-emulating oracle rollout (the band) and the negative branch of one CFG 4.5 forward pair: 10 oracle forwards at S = 4429, ~25 s each on the
GPU box's 128 host threads, i.e. most of the GPU suite's wall clock when computed inside the test.  The oracle is deterministic given its inputs,
so its outputs are committed instead and the test keeps only what depends on the ENGINE's output (the oracle replay of the engine's own transition).

The synthetic weights are drawn by the GPU generator (mi355_flow.weights.synthetic_state_dict(device="cuda", seed=1234): the CPU generator
needs ~1 min for 2.5 B values) -- so this script needs a GPU box although everything it computes runs on the host cores:

    gpurun -- 'python oracle/make_config_b_golden.py gpurun_out/config_b_oracle.npz'     # then copy to tests/golden/

Stored: the fixed spatial subsample [:, :, ::4, ::4] of every tensor the tests compare (16 x 32 x 32 of 16 x 128 x 128 latents: every 4 x 4 block
of the latent grid is sampled; rel-L2 over 16 384 elements estimates the full rel-L2 to ~1 %), fp16 / bf16 values stored exactly.
`MI355_CONFIG_B_LIVE=1` makes the test recompute everything in full instead of loading this file."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flow-factory_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_TEXT, STRIDE = 333, 4


def inputs():
    """The seeded inputs of the config-B tests (tests/test_gpu_fullsize.py: config_b)."""
    from oracle import rollout_ref as R, scheduler_ref as S
    B, h, w, N = 1, 128, 128, 4
    g = torch.Generator().manual_seed(4322)
    pe = torch.randn(B, N_TEXT, 4096, generator=g).bfloat16()
    pp = torch.randn(B, 2048, generator=g).bfloat16()
    ne = torch.randn(B, N_TEXT, 4096, generator=g).bfloat16()
    npl = torch.randn(B, 2048, generator=g).bfloat16()
    init, noise = R.draw_rollout_noise(B, 16, h, w, N, torch.bfloat16, torch.Generator().manual_seed(43))
    ts, sig = S.make_schedule(N, shift=3.0)
    sde = S.current_sde_steps([1, 2, 3], 1, 42, N)
    nl = S.noise_levels(N, sde, 0.7).tolist()
    return dict(B=B, h=h, w=w, N=N, pe=pe, pp=pp, ne=ne, npl=npl, init=init, noise=noise, ts=ts, sig=sig, nl=nl)


def compute(sd, cfg, c):
    """fp32 and bf16-emulating oracle rollouts + the negative branch of the CFG pair on the first state (full tensors)."""
    from oracle import mmditx_ref as M, rollout_ref as R, scheduler_ref as S
    with torch.no_grad():
        ref = R.rollout(sd, cfg, c["pe"], c["pp"], None, None, 1.0, c["init"], c["noise"], c["ts"], c["sig"], c["nl"], torch.float16)
        refq = R.rollout(sd, cfg, c["pe"], c["pp"], None, None, 1.0, c["init"], c["noise"], c["ts"], c["sig"], c["nl"], torch.float16, quant=M.bf16_round)
        x0 = S.cast_latents(c["init"], torch.float16)
        t_in = c["ts"][0].reshape(1).to(torch.float16).float()
        vu = M.mmdit_forward(sd, cfg, x0.float(), t_in, c["ne"].float(), c["npl"].float())
        vuq = M.mmdit_forward(sd, cfg, x0.float(), t_in, c["ne"].float(), c["npl"].float(), quant=M.bf16_round)
    return dict(lat=ref["all_latents"].float(), latq=refq["all_latents"].float(), lp=ref["log_probs"].float(), vt=ref["noise_preds"][0].float(),
                vtq=refq["noise_preds"][0].float(), vu=vu.float(), vuq=vuq.float())


def sub(x):
    return x[..., ::STRIDE, ::STRIDE].contiguous()


def main(out_path):
    if not torch.cuda.is_available():
        raise SystemExit("the synthetic weights of the full-size tests are drawn by the GPU generator: run this on a GPU box")
    from mi355_flow import engine
    from mi355_flow.weights import synthetic_state_dict
    from oracle import mmditx_ref as M
    sd_gpu = synthetic_state_dict(engine.TransformerConfig(), device="cuda", seed=1234, dtype=torch.bfloat16)
    sd = {k: v.float().cpu() for k, v in sd_gpu.items()}
    del sd_gpu
    c = inputs()
    o = compute(sd, M.SD35_MEDIUM, c)
    wsum = float(sum(float(v.double().sum()) for v in sd.values()))
    np.savez_compressed(out_path, stride=np.int64(STRIDE), n_text=np.int64(N_TEXT), weights_checksum=np.float64(wsum),
                        torch_version=np.array(torch.__version__), threads=np.int64(torch.get_num_threads()),
                        lat=sub(o["lat"]).half().numpy(), latq=sub(o["latq"]).half().numpy(), lp=o["lp"].numpy(),
                        vt=sub(o["vt"]).numpy(), vtq=sub(o["vtq"]).numpy(), vu=sub(o["vu"]).numpy(), vuq=sub(o["vuq"]).numpy())
    print("wrote", out_path, {k: tuple(v.shape) for k, v in o.items()}, "weights checksum", wsum)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "config_b_oracle.npz"))
