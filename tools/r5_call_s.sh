#!/bin/bash
# round 5 call S: merged backward launch -- lag sweep (3 repetitions) and fabric bytes (FETCH_SIZE / WRITE_SIZE) merged vs two launches
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5s; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in 1 2 3; do
for lag in 0 2 3 4 5 6 8; do
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python $R/tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$lag: /" >> $O/lag_sweep.txt
done
done
cat $O/lag_sweep.txt | sort | awk '{print $1, $(NF-6), $(NF-5)}' | tail -n 30
for m in 0 3; do
 for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$m$c
  MLA_ATTN_BWD_MERGED=$m timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$m$c -o p -- python $R/tools/bench_attn_step.py 548 32 > /dev/null 2>&1 < /dev/null
  echo "== merged=$m $c" >> $O/pmc_bytes.txt
  python $R/tools/pmc_summary.py $(find /tmp/pm_$m$c -name "*counter_collection.csv" | head -1) 'attn_(fwd|bwd)_\w+kernel' >> $O/pmc_bytes.txt 2>&1
 done
done
cat $O/pmc_bytes.txt
