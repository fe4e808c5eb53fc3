#!/bin/bash
# usage: tools/build_variants.sh name="-DFLAG=1 ..." ...   builds pecanpy_amd/lib_<name>.so for each (A/B runs: PECANPY_AMD_LIB)
cd "$(dirname "$0")/../pecanpy_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -fvisibility=hidden -Wno-unused-function"
pids=()
for spec in "$@"; do
  name="${spec%%=*}"; defs="${spec#*=}"
  ( /opt/rocm/bin/hipcc $FLAGS $defs -Rpass-analysis=kernel-resource-usage -o ../lib_$name.so pecanpy_amd.hip -Wl,-rpath,/opt/rocm/lib 2> /tmp/res_$name.txt \
      && echo "built lib_$name.so ($defs)" || { echo "FAILED $name"; grep -v remark /tmp/res_$name.txt | head -20; } ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
