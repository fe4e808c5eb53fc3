run() { timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('RMAT-22', d['value'], 'ms/pass', d['ms_per_step'], 'lane ms', r['avg_launch_ms'], 'entries/step', r['list_entries_per_step'], 'chain', r.get('float_chain_step_frac'), 'redo', r['redo_walks'])"; }
PW_DEBUG_ROUNDS=1 timeout 300 python tools/lanes_check.py 20 2>&1 | grep -E "p=|row" | head -60
echo "== main"; run
echo "== no queue"; PECANPY_AMD_NO_CHAIN_QUEUE=1 run
for tail in 20000 1000000; do echo "== tail $tail"; PECANPY_AMD_CHAIN_TAIL=$tail run; done
bash tools/ab_libs.sh lib_mw3.so lib_q5.so lib_c256.so
PW_DEBUG_ROUNDS=1 timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep round
PECANPY_AMD_LIB=$PWD/pecanpy_amd/lib_pq.so timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "lane_prof"
