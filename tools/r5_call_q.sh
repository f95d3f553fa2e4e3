#!/bin/bash
# round 5 call Q: front-end timeline of one traced configs[1] step (what runs before the first decoder-layer attention)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5q; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_q
timeout 900 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_q -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > $O/bench.json 2> $O/bench.err < /dev/null
DB=$(find /tmp/prof_q -name "*.db" | head -1)
if [ -n "$DB" ]; then timeout 300 python $R/tools/step_frontend_timeline.py $DB 1 > $O/frontend_timeline.txt 2>&1; fi
tail -n 5 $O/frontend_timeline.txt; cut -c1-200 $O/bench.json
