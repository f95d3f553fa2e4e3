#!/bin/bash
# round 5, GPU call I: in-step per-kernel attention times, swizzle 0 vs swizzle 1 builds, same box, alternating (rocprofv3 kernel stats of 4 steps)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5i; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for r in 1 2; do
  for t in sw0 sw1p; do
    lib=$R/mla_amd/csrc/build_exp/sw0/libmla_hip.so; [ $t = sw1p ] && lib=$R/mla_amd/libmla_hip.so
    rm -rf /tmp/pi_$t$r
    MLA_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi_$t$r -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > $O/bench_$t$r.json 2>/dev/null < /dev/null
    f=$(find /tmp/pi_$t$r -name "*kernel_stats.csv" 2>/dev/null | head -1)
    echo "== $t run $r: $(python -c "import json;print(json.loads(open('$O/bench_$t$r.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>&1 | tail -1) ms/step" | tee -a $O/attn_instep.txt
    [ -n "$f" ] && python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'attn_' in r['Name'] and 'local' not in r['Name']: print('   %-40s calls %s avg %.1f us' % (r['Name'].split('::')[1][:40], r['Calls'], float(r['AverageNs'])/1e3))
" | tee -a $O/attn_instep.txt
  done
done
