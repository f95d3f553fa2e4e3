#!/bin/bash
# round 5, GPU call J: N-rank rehearsal of bench.py after this round's edits + the new true-dimension point-cloud head test
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_generation_gpu.py -q -s -k "true_dimensions or 7b_dimensions or heads_against" > $O/test_gen.txt 2>&1 < /dev/null; echo "gen rc=$?"; tail -n 4 $O/test_gen.txt
timeout 1200 bash tools/rehearse_bench_ranks.sh 2 8 > $O/rehearsal.txt 2>&1 < /dev/null; echo "rehearsal rc=$?"; grep -v "^\[W\|Warning\|warn" $O/rehearsal.txt | tail -n 16
