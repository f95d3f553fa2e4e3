#!/bin/bash
# Round evidence in one GPU call: bench lines of configs 1 / 3 / 4, rocprofv3 kernel stats of the same config-1 command, and the two
# PMC passes behind roofline.traffic, and the in-step per-kernel breakdown (tools/step_breakdown.py: one step, init excluded). Usage (GPU box): bash tools/collect_profiles.sh <tag>   ->   gpurun_out/<tag>_*
TAG=${1:-r2}
R=/root/repo; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 8 --warmup 2 > $O/${TAG}_bench_config1.json 2> $O/${TAG}_bench_config1.err
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_$TAG -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_config1_profiled.json 2>/dev/null
python $R/tools/step_breakdown.py $(find /tmp/prof_$TAG -name "*.db" | head -1) 1 60 > $O/${TAG}_step_breakdown_config1.txt
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench7b_config1_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-gemm-profile > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_gemm256_$c.txt
done
python $R/tools/pmc_reduce.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/${TAG}_gemm256_hbm_traffic.json
for cfg in 3 4; do
  python $R/bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_config$cfg.json 2>/dev/null
  rm -rf /tmp/prof_c$cfg
  rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_c$cfg -o p -- python $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  cp $(find /tmp/prof_c$cfg -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench7b_config${cfg}_kernel_stats.csv
  python $R/tools/step_breakdown.py $(find /tmp/prof_c$cfg -name "*.db" | head -1) 1 40 > $O/${TAG}_step_breakdown_config$cfg.txt
done
ls -la $O | grep ${TAG}_
