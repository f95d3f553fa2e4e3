#!/bin/bash
# Bench lines + kernel stats + in-step breakdowns of BASELINE configs[3] and [4] (and the headline config once more, with cpu_baseline).
TAG=${1:-r3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 8 --warmup 2 > $O/${TAG}_bench_config1.json 2> $O/${TAG}_bench_config1.err
for cfg in 3 4; do
  python $R/bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_config$cfg.json 2>/dev/null
  rm -rf /tmp/prof_c$cfg
  rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_c$cfg -o p -- python $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  cp $(find /tmp/prof_c$cfg -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench7b_config${cfg}_kernel_stats.csv
  python $R/tools/step_breakdown.py $(find /tmp/prof_c$cfg -name "*.db" | head -1) 1 40 > $O/${TAG}_step_breakdown_config$cfg.txt
done
# configs[4] above = the reference's policy (every layer checkpointed, the default since round 4); the opt-in mixed policy sized from measured memory:
python $R/bench.py --config 4 --keep-layers -1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_config4_mixed_policy.json 2> $O/${TAG}_bench_config4_mixed_policy.err
for f in config1 config3 config4 config4_mixed_policy; do python -c "import json;d=json.load(open('$O/${TAG}_bench_$f.json'));print('$f',d['ms_per_step'],d['value'],d['peak_mem_gb'],d['config'].get('activation_policy',''))"; done
