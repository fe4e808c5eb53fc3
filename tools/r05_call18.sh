#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c18; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_lanes.py -m gpu -x -q -k "weighted" 2>&1 | tail -2
timeout 500 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "c5" 2>&1 | tail -2
timeout 300 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-host-call 2> $O/C5.err | tail -1 > $O/C5.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c18/C5.json")); r = d["roofline"]
print("C5", d["value"], d["ms_per_step"], "eager frac", r.get("eager_step_frac"), "rounds", r.get("lane_rounds"), "param ms", d["config"]["param_index_build_ms"])
PY
