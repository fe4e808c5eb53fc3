#!/bin/bash
# usage: tools/prof_sections.sh [bench args...]      (run on the GPU box, e.g. through gpurun)
# Builds a -DPW_PROF variant of the library (per-section s_memtime accounting inside walk_kernel, see
# walk_sparse.hip.h: Prof) next to the regular one and runs bench.py with it; the section shares and
# per-step event counts are printed on stderr as [pw_prof] lines.  Timing with this build is ~10 % slower.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SO=$R/pecanpy_amd/ab_prof.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared \
    -fvisibility=hidden -DPW_PROF -o "$SO" -x hip "$R/pecanpy_amd/csrc/pecanpy_amd.hip" -Wl,-rpath,/opt/rocm/lib
PECANPY_AMD_LIB=$SO python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" 2>&1 | grep -E "pw_prof|metric" | cut -c1-200
