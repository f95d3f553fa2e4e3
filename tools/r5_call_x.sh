#!/bin/bash
# round 5 call X: profiles of record on the final build (merged backward launch): kernel stats + step breakdown + roofline table + PMC
# table + traffic records + configs 3 / 4 (tools/collect_counters.sh, tools/collect_profiles.sh), tag r5f
R=${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=$R
timeout 1500 bash $R/tools/collect_counters.sh r5f > /dev/null 2>&1
timeout 2400 bash $R/tools/collect_profiles.sh r5f > /dev/null 2>&1
ls -la $R/gpurun_out | grep r5f_ | awk '{print $5, $9}'
head -n 12 $R/gpurun_out/r5f_step_breakdown_config1.txt | cut -c1-120
grep -i "attn" $R/gpurun_out/r5f_pmc_table.txt | cut -c1-250
