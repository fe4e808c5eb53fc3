#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
cat > /tmp/create.py <<'PY'
import numpy as np, sys, time
import os; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] if "GRAFT_REPO_ROOT" in os.environ else "/root/repo")
from pecanpy_amd.synth import rmat_csr
from pecanpy_amd.engine import WalkEngine
import torch
torch.cuda.init(); torch.cuda.synchronize()
indptr, indices, data = rmat_csr(22, seed=1)
t = time.perf_counter(); eng = WalkEngine.from_csr(indptr, indices, None); print("create wall ms", (time.perf_counter() - t) * 1e3, eng.index_info())
PY
cd /tmp; rm -rf /tmp/kc
PECANPY_AMD_CREATE_DEBUG=1 rocprofv3 --kernel-trace --stats -d /tmp/kc -o p -- python /tmp/create.py 2>&1 | grep -E "create|lane index" | cut -c1-160
cd $R; python tools/prof_summary.py /tmp/kc/p_results.db /tmp/kc.txt | head -30 | cut -c1-150
