#!/bin/bash
# round 5, GPU call H: final build -- whole -m gpu suite, lga_prep_bwd timing, counters / profiles of record (tag r5)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
bash tools/run_gpu_tests.sh > $O/r5_gputest_summary.txt 2>&1; cp $O/gputest.log $O/r5_gpu_test_log.txt; head -3 $O/r5_gputest_summary.txt
python tools/bench_lga_prep_bwd.py > $O/r5_lga_prep_bwd_timing.txt 2>&1; cat $O/r5_lga_prep_bwd_timing.txt | grep stage
bash tools/collect_counters.sh r5 all > $O/r5_collect.log 2>&1; tail -6 $O/r5_collect.log; cat $O/r5_pmc_table.txt | tail -10
