#!/bin/bash
# round 5 call V: merged backward launch as the default -- whole GPU suite, smoke, stand-alone timing, default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5v; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 < /dev/null | tail -n 15 > $O/gpu_test_log.txt; tail -n 3 $O/gpu_test_log.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | tail -n 2 | tee $O/smoke.txt
for m in 0 105 0 105; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$m: /" | tee -a $O/timing.txt
done
( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err < /dev/null; tail -n 4 $O/bench_default.err; cut -c1-300 $O/bench_default.json
