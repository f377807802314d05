"""Per-kernel timing of the hot ops at SD3.5-medium 1024^2 shapes (prints TFLOP/s / GB/s)."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import engine  # noqa: E402


def timeit(fn, iters=10, warm=3):
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


def main():
    B = int(os.environ.get("MB_BATCH", "8"))
    dev = "cuda"
    print(f"batch {B}")
    for (M, N, K, name) in [(B * 4096, 3072, 1536, "qk_img"), (B * 4096, 1536, 1536, "out_img"),
                            (B * 4096, 6144, 1536, "ff1_img"), (B * 4096, 1536, 6144, "ff2_img"),
                            (B * 333, 3072, 1536, "qk_ctx"), (B * 333, 6144, 1536, "ff1_ctx"), (B * 333, 1536, 6144, "ff2_ctx"),
                            (8192, 8192, 8192, "square8k"), (4096, 4096, 4096, "square4k")]:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        b = torch.zeros(N, device=dev)
        t = timeit(lambda: engine.op_linear(x, w, b, 0))
        print(f"gemm {name:10s} M={M:6d} N={N:5d} K={K:5d}: {t*1e3:8.3f} ms  {2.0*M*N*K/t/1e12:8.1f} TFLOP/s")
        tt = timeit(lambda: torch.nn.functional.linear(x, w))
        print(f"     torch/hipBLASLt reference                : {tt*1e3:8.3f} ms  {2.0*M*N*K/tt/1e12:8.1f} TFLOP/s")
    for (H, S, n_img, name) in [(24, 4429, 4096, "joint"), (24, 4096, 4096, "dual")]:
        S_pad = (S + 63) // 64 * 64
        q = torch.randn(B, H, S_pad, 64, device=dev).bfloat16()
        k = torch.randn(B, H, S_pad, 64, device=dev).bfloat16()
        vT = torch.randn(B, H, 64, S_pad, device=dev).bfloat16()
        t = timeit(lambda: engine.op_attention(q, k, vT, S, n_img))
        fl = 4.0 * B * H * S * S * 64
        print(f"attn {name:6s} B={B} H={H} S={S}: {t*1e3:8.3f} ms  {fl/t/1e12:8.1f} TFLOP/s")
        qq, kk, vv = q[:, :, :S], k[:, :, :S], vT.transpose(2, 3)[:, :, :S]
        tt = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qq, kk, vv))
        print(f"     torch SDPA reference      : {tt*1e3:8.3f} ms  {fl/tt/1e12:8.1f} TFLOP/s")
    M, D = B * 4096, 1536
    x = torch.randn(M, D, device=dev).bfloat16()
    sh = torch.randn(B, D, device=dev).bfloat16()
    t = timeit(lambda: engine.op_ln_modulate(x, sh, sh, 4096))
    print(f"ln_mod M={M}: {t*1e3:.3f} ms  {2.0*M*D*2/t/1e9:.0f} GB/s (incl. wrapper overhead)")


if __name__ == "__main__":
    main()
