#!/bin/bash
# PMC: cache / fetch / issue counters of the lane kernels for two builds (one pass each, ab_bench child under rocprofv3)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out/c3
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python tools/ab_bench.py --passes 0 libpecanpy_amd.so > /dev/null 2>&1   # (generates /tmp/ab_rmat22.npz)
cd /tmp
for lib in libpecanpy_amd.so lib_drawlds.so lib_base.so; do
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    rm -rf /tmp/pq
    PECANPY_AMD_LIB=$R/pecanpy_amd/$lib timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pq -o p -- python $R/tools/ab_bench.py --child --graph /tmp/ab_rmat22.npz --passes 0 > /tmp/pq.log 2>&1
    echo "== $lib group $i" >> $R/gpurun_out/c3/pmc.txt
    python $R/tools/prof_summary.py /tmp/pq/p_results.db 2>/dev/null | grep -E "walk_lanes|lanes_chain" >> $R/gpurun_out/c3/pmc.txt
  done
done
cat $R/gpurun_out/c3/pmc.txt
