"""s_memtime trace of the persistent ping-pong GEMM: per workgroup and tile {tile start, main loop start, main loop end, stores drained}.
s_memtime ticks at a CONSTANT 2.4 GHz on MI355X whatever the shader clock is (scripts/mb/clock_calib.hip: 2402.8 MHz idle, 2398 MHz under
MFMA load, against the 100 MHz s_memrealtime) -- ticks are time (0.4167 ns), NOT core cycles."""
import math, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import _lib
lib = _lib.load()
TICK_US = 1.0 / 2400.0
for (M, N, K) in [(32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.zeros(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tr = torch.zeros(256 * 16 * 2 * 4, device="cuda", dtype=torch.int64)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.mi355_op_linear_trace(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, tr.data_ptr())
    torch.cuda.synchronize()
    tr.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.mi355_op_linear_trace(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, tr.data_ptr())
    e1.record(); torch.cuda.synchronize()
    ev_us = e0.elapsed_time(e1) * 1e3
    t = tr.cpu().numpy().reshape(256, 16, 2, 4).astype(np.float64)
    ntile = int((t[0, :15, 0, 3] > 0).sum())          # slot 15 holds the device-wide stamps, not a tile
    starts = t[:, 0, 0, 0]; ends = t[:, ntile - 1, :, 3].max(axis=1)
    live = starts > 0
    t0 = starts[live].min()
    s_us, e_us = (starts[live] - t0) * TICK_US, (ends[live] - t0) * TICK_US
    print(f"M={M} N={N} K={K}: {ntile} tiles/WG, {int(live.sum())} WGs | hipEvent {ev_us:.1f} us | WG start after first: p50 {np.median(s_us):.1f} p90 {np.percentile(s_us,90):.1f} max {s_us.max():.1f} us"
          f" | WG end: min {e_us.min():.1f} p50 {np.median(e_us):.1f} max {e_us.max():.1f} us | per-WG busy p50 {np.median(e_us - s_us):.1f} us")
    ml = (t[live][:, :ntile, :, 2] - t[live][:, :ntile, :, 1]) * TICK_US
    ep = (t[live][:, :ntile, :, 3] - t[live][:, :ntile, :, 2]) * TICK_US
    print(f"   main loop per tile p50 {np.median(ml):.2f} us (min {ml.min():.2f} max {ml.max():.2f}) = {np.median(ml) / (K // 64) * 1e3:.0f} ns per K-tile; epilogue+drain p50 {np.median(ep):.2f} us")
    for xcd in range(8):
        m = live & (np.arange(256) % 8 == xcd)
        print(f"   XCD{xcd}: start p50 {np.median((starts[m]-t0)*TICK_US):6.1f}  end p50 {np.median((ends[m]-t0)*TICK_US):6.1f} max {((ends[m]-t0)*TICK_US).max():6.1f}")
    ws, we = t[live][:, 15, 0, 0] * 0.01, t[live][:, 15, :, 3].max(axis=1) * 0.01       # s_memrealtime: 10 ns ticks, device-wide time base
    w0 = ws.min()
    print(f"   device-wide clock: WG starts p50 {np.median(ws - w0):.1f} p90 {np.percentile(ws - w0, 90):.1f} max {(ws - w0).max():.1f} us after the first; "
          f"WG ends min {(we - w0).min():.1f} p50 {np.median(we - w0):.1f} max {(we - w0).max():.1f} us; sorted starts: " + " ".join(f"{v:.0f}" for v in np.sort(ws - w0)[::16]))
    # dispatch stagger inside one XCD (its s_memtime counter is common to its CUs): sorted start / end offsets from the XCD's first start
    m = live & (np.arange(256) % 8 == 3)
    s3 = np.sort((starts[m] - starts[m].min()) * TICK_US); e3 = np.sort((ends[m] - starts[m].min()) * TICK_US)
    print("   XCD3 WG starts (us after the XCD's first):", " ".join(f"{v:.0f}" for v in s3))
    print("   XCD3 WG ends   (us after the XCD's first):", " ".join(f"{v:.0f}" for v in e3))
