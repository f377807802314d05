"""Core clock DELIVERED to the hot kernels under the package power cap, and what their roofline fractions are against THAT clock.

One probe wave (mi355_clock_probe, csrc/elementwise.hip) on a side stream samples {s_memtime = core clocks, s_memrealtime = 100 MHz wall
clock} every few microseconds while the main stream loops one kernel; delivered MHz = d(core clocks) / d(wall).  The nominal peaks (2.5 PF
bf16 MFMA) are quoted at 2.4 GHz; at the ~1.35 kW cap the bf16 GEMM was seen at an effective 1.44 GHz (profiles/r02_power_clock_notes.txt).
Per kernel this prints: TFLOP/s, delivered MHz (p10 / median / p90), the MFMA ceiling at the delivered clock, and -- for the d = 64
attention -- the v_exp_f32 issue ceiling at that clock (32 quarter-rate v_exp + 16 v_cvt_pk + 16 v_dot2c + ~10 other VALU per 16 MFMA:
~720 VALU cycles against 512 MFMA cycles per 64-key tile and wave; ISA of attn_kernel<1,8,1,0>).

NOT YET RUN ON THE GPU (written with no GPU time left in round 2).   usage: python scripts/clock_under_load.py [--ms 40]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import _lib, engine  # noqa: E402

lib = _lib.load()
NOMINAL_MHZ, PEAK_TF = 2400.0, 2500.0
ATTN64_VALU_CYC, ATTN64_MFMA_CYC = 720.0, 512.0       # per wave and 64-key tile (static-softmax kernel)


def per_launch_s(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def probe(fn, span_ms, side):
    """Loop fn for ~2 * span_ms on the current stream, sample the clocks for ~span_ms from a side stream in the middle of it."""
    t1 = per_launch_s(fn) if fn is not None else None
    n = 2000
    out = torch.zeros(2 * n, device="cuda", dtype=torch.int64)
    # ~8.1 k core clocks per s_sleep(127): aim for span_ms at a ~1.8 GHz clock
    sleep_iters = max(1, int(span_ms * 1e-3 * 1.8e9 / 8128 / n))
    torch.cuda.synchronize()
    if fn is not None:
        reps = max(3, int(math.ceil(2.0 * span_ms * 1e-3 / t1)))
        for _ in range(max(1, reps // 4)):
            fn()                                              # the load is running before the probe starts
        _lib.check(lib.mi355_clock_probe(side.cuda_stream, out.data_ptr(), n, sleep_iters), "clock_probe")
        for _ in range(reps):
            fn()
    else:
        _lib.check(lib.mi355_clock_probe(side.cuda_stream, out.data_ptr(), n, sleep_iters), "clock_probe")
    torch.cuda.synchronize()
    a = out.cpu().numpy().reshape(n, 2).astype(np.float64)
    dc, dw = np.diff(a[:, 0]), np.diff(a[:, 1])
    ok = dw > 0
    mhz = dc[ok] / dw[ok] * 100.0                             # s_memrealtime ticks at 100 MHz
    span = (a[-1, 1] - a[0, 1]) / 100.0 * 1e-3                # ms
    return t1, np.percentile(mhz, 10), np.median(mhz), np.percentile(mhz, 90), span


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", type=float, default=40.0, help="probe span per kernel")
    args = ap.parse_args()
    side = torch.cuda.Stream()
    rows = []
    _, p10, med, p90, span = probe(None, 5.0, side)
    print(f"idle device: {med:.0f} MHz (p10 {p10:.0f}, p90 {p90:.0f}) over {span:.1f} ms", flush=True)

    def report(name, flops, fn, valu_bound=None):
        t1, p10, med, p90, span = probe(fn, args.ms, side)
        tf = flops / t1 / 1e12
        ceil = PEAK_TF * med / NOMINAL_MHZ
        line = (f"{name:44s} {t1 * 1e6:9.1f} us {tf:7.1f} TF | delivered {med:5.0f} MHz (p10 {p10:.0f}, p90 {p90:.0f}; {span:.0f} ms) | "
                f"{tf / PEAK_TF:5.1%} of nominal, {tf / ceil:5.1%} of the MFMA ceiling at this clock ({ceil:.0f} TF)")
        if valu_bound is not None:
            vb = ceil * valu_bound
            line += f" | {tf / vb:5.1%} of the v_exp issue ceiling ({vb:.0f} TF)"
        print(line, flush=True)
        rows.append(line)

    # ---- GEMMs of the image stream at the bench shape (forward batch 8 at 1024^2: M = 32768)
    for (M, N, K) in ((32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144), (32768, 3072, 1536)):
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
        b = torch.zeros(N, device="cuda")
        report(f"gemm_pp {M}x{N}x{K}", 2.0 * M * N * K, lambda: engine.op_linear(x, w, b, 0))
        wb = w.clone()
        report(f"hipBLASLt {M}x{N}x{K}", 2.0 * M * N * K, lambda: torch.nn.functional.linear(x, wb))
        del x, w, b, wb

    # ---- attention, head_dim 64 (joint S = 4429 and dual S = 4096 of SD3.5-medium at 1024^2, B = 8)
    B, H = 8, 24
    for (S, n_img) in ((4429, 4096), (4096, 4096)):
        S_pad = (S + 63) // 64 * 64
        q = torch.zeros(B, H, S_pad, 64, device="cuda", dtype=torch.bfloat16)
        k = torch.zeros_like(q)
        v = torch.zeros_like(q)
        q[:, :, :S] = torch.randn(B, H, S, 64, device="cuda").bfloat16()
        k[:, :, :S] = torch.randn(B, H, S, 64, device="cuda").bfloat16()
        v[:, :, :S] = torch.randn(B, H, S, 64, device="cuda").bfloat16()
        vT = v.transpose(2, 3).contiguous()
        fl = 4.0 * B * H * S * S * 64
        for static in (1, 0):
            lib.mi355_tune_set(6, 40 if static else 0)        # op-level entry: key 6 >= 2 is the proven |score| bound
            report(f"attention d=64 S={S} {'static' if static else 'dynamic'} softmax", fl, lambda: engine.op_attention(q, k, vT, S, n_img),
                   valu_bound=ATTN64_MFMA_CYC / ATTN64_VALU_CYC if static else None)
        lib.mi355_tune_set(6, 1)
        del q, k, v, vT

    # ---- a registers-only MFMA loop would show the unthrottled clock; the LayerNorm-modulate kernel shows an HBM-bound one
    M, D = 32768, 1536
    x = torch.randn(M, D, device="cuda").bfloat16()
    sh = torch.randn(8, D, device="cuda").bfloat16()
    sc = torch.randn(8, D, device="cuda").bfloat16()
    if hasattr(engine, "op_ln_modulate"):
        t1, p10, med, p90, span = probe(lambda: engine.op_ln_modulate(x, sh, sc, M // 8), args.ms, side)
        print(f"{'ln_modulate 32768x1536 (HBM-bound)':44s} {t1 * 1e6:9.1f} us {2.0 * M * D * 2 / t1 / 1e12:7.2f} TB/s | delivered {med:5.0f} MHz "
              f"(p10 {p10:.0f}, p90 {p90:.0f})", flush=True)


if __name__ == "__main__":
    main()
