cd $GRAFT_REPO_ROOT
python - <<'PY' &
import math, os, sys, torch, time
sys.path.insert(0, "flow-factory_amd")
from mi355_flow import engine
M,N,K=32768,1536,1536
x=torch.randn(M,K,device="cuda").bfloat16(); w=(torch.randn(N,K,device="cuda")/math.sqrt(K)).bfloat16(); b=torch.zeros(N,device="cuda")
t0=time.time()
while time.time()-t0<6: 
    for _ in range(200): engine.op_linear(x,w,b,0)
    torch.cuda.synchronize()
print("gemm loop done")
t0=time.time()
while time.time()-t0<6:
    for _ in range(200): torch.nn.functional.linear(x,w)
    torch.cuda.synchronize()
print("hipblaslt loop done")
PY
sleep 3
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4; sleep 1; done
echo ---- hipblaslt phase
sleep 3
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4; sleep 1; done
wait
