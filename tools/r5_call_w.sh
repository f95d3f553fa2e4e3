#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5w; mkdir -p $O
timeout 3000 python -X faulthandler -m pytest tests/ -v -m gpu > $O/gpu_test_full.txt 2>&1 < /dev/null
grep -c "PASSED" $O/gpu_test_full.txt; grep -n "FAILED\|ERROR" $O/gpu_test_full.txt | head -n 10; tail -n 3 $O/gpu_test_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | tail -n 1
