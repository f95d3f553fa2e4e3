#!/bin/bash
# Same-box A/B of whole configs[1] steps between environment settings: tools/ab_step.sh "<env A>" "<env B>" ... (two alternating rounds)
for r in 1 2; do for e in "$@"; do
  ms=$(env $e python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "[$e] $ms ms/step"
done; done
