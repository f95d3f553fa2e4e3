#!/bin/bash
# round 5 call P: the round-end sequence on the final commit -- whole GPU suite, smoke, the default bench line (with `secondary`),
# then the 120-step soak.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5p2; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 < /dev/null | tail -n 15 > $O/gpu_test_log.txt; tail -n 3 $O/gpu_test_log.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | tail -n 3 | tee $O/smoke.txt
( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err < /dev/null; tail -n 4 $O/bench_default.err; cut -c1-400 $O/bench_default.json
timeout 900 python tools/soak.py 120 2>&1 < /dev/null | grep -v amdgpu > $O/soak_120steps.txt; tail -n 3 $O/soak_120steps.txt
