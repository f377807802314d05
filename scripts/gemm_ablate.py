"""Ablations of the ping-pong GEMM main loop (results are garbage, timings are not): what each resource costs at the power cap."""
import math, os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
    import torch
    from mi355_flow import _lib
    lib = _lib.load()
    out = {}
    for (M, N, K) in [(32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144)]:
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
        b = torch.zeros(N, device="cuda"); o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        st = torch.cuda.current_stream().cuda_stream
        f = lambda: lib.mi355_op_linear_trace(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), M, N, K, None)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        out[f"{M}x{N}x{K}"] = round(2.0 * M * N * K / (e0.elapsed_time(e1) / 30 * 1e-3) / 1e12, 1)
    print(json.dumps(out))
else:
    for mask, what in ((0, "shipped"), (1, "no K-loop prefetch (global_load_lds)"), (2, "no LDS fragment reads"), (3, "neither: MFMA + barriers + epilogue"),
                       (8, "epilogue without its global stores"), (4, "no epilogue at all"), (7, "MFMA + barriers only")):
        env = dict(os.environ, MI355_DBG_MASK=str(mask))
        r = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True)
        print(f"mask {mask} {what:45s} TFLOP/s-equivalent: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}")
