#!/bin/bash
# Box lottery guard: the same build spans +-2.5 % across boxes (board power cap). Run the evidence collection only on a box that is not
# one of the slow ones. Usage (GPU box): bash tools/collect_if_typical.sh <tag> <max_ms>
TAG=${1:-r2}; MAXMS=${2:-638}
ms=$(python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "probe: $ms ms/step (limit $MAXMS)"
if python -c "import sys; sys.exit(0 if float('$ms') <= float('$MAXMS') else 1)"; then
  bash /root/repo/tools/collect_profiles.sh $TAG
else
  echo "slow box: collection skipped"
fi
