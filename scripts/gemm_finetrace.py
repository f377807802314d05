import math, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-factory_amd"))
from mi355_flow import _lib
lib = _lib.load()
M, N, K = 32768, 1536, 1536
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
b = torch.zeros(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
tr = torch.zeros(65536 + 256 * 2 * 32, device="cuda", dtype=torch.int64)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    tr.zero_()
    lib.mi355_op_linear_trace(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, tr.data_ptr())
torch.cuda.synchronize()
ft = tr[65536:].cpu().numpy().reshape(256, 2, 32).astype(np.float64)
names = ["P0 read+glds issue", "P0 vmcnt wait", "P0 barrier1", "P0 MFMA(16)", "P0 barrier2 ", "P1 read+glds issue", "P1 vmcnt wait", "P1 barrier1", "P1 MFMA", "P1 barrier2",
         "P2 read+glds", "P2 barrier1", "P2 MFMA", "P2 barrier2", "P3 glds", "P3 vmcnt wait", "P3 barrier1", "P3 MFMA", "P3 barrier2"]
for g in (0, 1):
    d = np.diff(ft[:, g, :20], axis=1)
    print(f"group {g}: step total {np.median(ft[:, g, 19] - ft[:, g, 0]):.0f} cycles")
    for i, n in enumerate(names):
        print(f"   {n:22s} median {np.median(d[:, i]):6.0f}  p90 {np.percentile(d[:, i], 90):6.0f}")
