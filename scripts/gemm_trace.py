import math, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import _lib
lib = _lib.load()
for (M, N, K) in [(32768, 1536, 1536), (32768, 6144, 1536), (32768, 1536, 6144)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.zeros(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tr = torch.zeros(256 * 16 * 2 * 4, device="cuda", dtype=torch.int64)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        tr.zero_()
        lib.mi355_op_linear_trace(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, tr.data_ptr())
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(256, 16, 2, 4).astype(np.float64)
    ntile = int((t[0, :, 0, 3] > 0).sum())
    t0 = t[:, 0, 0, 0].min()
    print(f"M={M} N={N} K={K}: tiles/wg={ntile}  (s_memtime ticks, 100 MHz => 10 ns each)")
    for g in (0, 1):
        for ti in range(ntile):
            s = t[:, ti, g, :]
            print(f"  grp{g} tile{ti}: start@{np.median(s[:,0]-t0):8.0f} wait {np.median(s[:,1]-s[:,0]):6.0f}  mainloop {np.median(s[:,2]-s[:,1]):6.0f}  epilogue+drain {np.median(s[:,3]-s[:,2]):6.0f}   (min/max mainloop {np.min(s[:,2]-s[:,1]):.0f}/{np.max(s[:,2]-s[:,1]):.0f})")
    print(f"  kernel span {t[:, :ntile, :, 3].max() - t0:.0f} ticks")
