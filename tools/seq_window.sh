#!/bin/bash
# Launch sequence of a window of one configs[1] step (name, duration, gap in front), from a rocprofv3 kernel trace:
#   tools/seq_window.sh <from_ms> <n launches>      (from_ms = offset inside the step; ~190 = end of the forward, ~300 = mid-backward)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_seq -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-gemm-profile > /dev/null 2>&1
SEQ=${2:-80} SEQ_FROM_MS=${1:-300} python $GRAFT_REPO_ROOT/tools/step_breakdown.py $(find /tmp/prof_seq -name "*.db" | head -1) 1 5 | tail -$((${2:-80} + 3))
