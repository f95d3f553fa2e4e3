#!/bin/bash
# Counter evidence for north_star's "rocprof HBM GB/s and MFMA utilisation against chip peak" (VERDICT r2 item 6). One GPU call:
#   (a) kernel-trace stats of the bench command            -> <tag>_bench7b_config1_kernel_stats.csv, <tag>_step_breakdown_config1.txt
#   (b) PMC pass 1 (SQ + GRBM): MFMA-busy, VALU-active, LDS bank conflicts / index cycles, wave cycles, waits
#       PMC pass 2 (TCC): L2 hits / misses; passes 3, 4: FETCH_SIZE, WRITE_SIZE (one counter set per run, --pmc never mixed with tracing domains)
#   (c) rocm-smi power + sclk trace sampled every 0.2 s while bench.py runs
# The reductions (tools/pmc_table.py, tools/roofline_table.py, tools/power_summary.py) run in the same call; everything lands in gpurun_out/.
TAG=${1:-r3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# (c) first: un-profiled run with the sampler beside it
( for i in $(seq 1 400); do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done ) > $O/${TAG}_power_samples.jsonl &
SP=$!
python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_config1_power_run.json 2> /dev/null
kill $SP 2>/dev/null
python $R/tools/power_summary.py $O/${TAG}_power_samples.jsonl $O/${TAG}_bench_config1_power_run.json > $O/${TAG}_power_trace.txt
# (a)
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_$TAG -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_config1_profiled.json 2>/dev/null
python $R/tools/step_breakdown.py $(find /tmp/prof_$TAG -name "*.db" | head -1) 1 60 > $O/${TAG}_step_breakdown_config1.txt
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench7b_config1_kernel_stats.csv
python $R/tools/roofline_table.py $O/${TAG}_step_breakdown_config1.txt > $O/${TAG}_roofline_table.txt
# (b)
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-gemm-profile --no-box > /dev/null 2>&1
  cp $(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1) /tmp/pmc_$i.csv
done
python $R/tools/pmc_table.py /tmp/pmc_1.csv /tmp/pmc_2.csv /tmp/pmc_3.csv /tmp/pmc_4.csv $O/${TAG}_pmc_table.json 2 > $O/${TAG}_pmc_table.txt
python $R/tools/pmc_reduce.py /tmp/pmc_3.csv /tmp/pmc_4.csv $O/${TAG}_gemm256_hbm_traffic.json 150 /tmp/pmc_2.csv
cat $O/${TAG}_power_trace.txt | tail -12; cat $O/${TAG}_pmc_table.txt; head -40 $O/${TAG}_roofline_table.txt
# second argument "all": the secondary configurations from the same command (bench lines of configs[3] / [4] with their own `roofline`
# objects, kernel stats, step breakdowns; configs[4] with the reference's policy and with the opt-in mixed policy)
if [ "$2" = "all" ]; then bash $R/tools/collect_configs34.sh $TAG; fi
