#!/bin/bash
# Round 5, call 6: smoke() in the three ways it can be reached (call 5: `python __graft_entry__.py smoke` -- build() then smoke() in one process,
# the library loaded before torch -- failed its first hipMalloc; the driver's own form, smoke() alone, is the one that ran green in rounds 1-4).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05f; mkdir -p $O
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__ (smoke alone)')" 2>&1 | tail -n 3 ) > $O/smoke_order.txt
( timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('__SMOKE_OK__ (build then smoke, one process)')" 2>&1 | tail -n 3 ) >> $O/smoke_order.txt
( timeout 600 python __graft_entry__.py smoke 2>&1 | tail -n 3 ) >> $O/smoke_order.txt
( timeout 300 python - <<'PY' 2>&1 | tail -n 4
import ctypes, os
lib = ctypes.CDLL(os.path.join("flow-factory_amd", "libmi355flow.so"))      # the library BEFORE torch, by hand
import torch
print("lib first, then torch: cuda available", torch.cuda.is_available(), "| tensor on gpu", float(torch.ones(4, device="cuda").sum()))
PY
) >> $O/smoke_order.txt
cat $O/smoke_order.txt
