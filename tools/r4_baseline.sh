#!/bin/bash
# Round-4 baseline on a fresh box: un-profiled bench line + per-kernel step breakdown of configs[1] (no counters).
TAG=${1:-r4a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_config1.json 2> $O/${TAG}_bench_err.txt
tail -1 $O/${TAG}_bench_config1.json | cut -c1-600
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_$TAG -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_config1_profiled.json 2>/dev/null
python $R/tools/step_breakdown.py $(find /tmp/prof_$TAG -name "*.db" | head -1) 1 60 > $O/${TAG}_step_breakdown_config1.txt
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench7b_config1_kernel_stats.csv
python $R/tools/roofline_table.py $O/${TAG}_step_breakdown_config1.txt > $O/${TAG}_roofline_table.txt
head -60 $O/${TAG}_step_breakdown_config1.txt
