#!/bin/bash
# round 5 call U: whole-step same-box A/B of the merged backward launch: configs[1] (3 alternating rounds) and configs[4]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5u; mkdir -p $O
for r in 1 2 3; do for m in 0 105; do
  ms=$(MLA_ATTN_BWD_MERGED=$m timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile 2>/dev/null < /dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "configs[1] MLA_ATTN_BWD_MERGED=$m: $ms ms/step" | tee -a $O/step_ab.txt
done; done
for r in 1 2; do for m in 0 104; do
  ms=$(MLA_ATTN_BWD_MERGED=$m timeout 900 python bench.py --config 4 --keep-layers 0 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-gemm-profile 2>/dev/null < /dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "configs[4] MLA_ATTN_BWD_MERGED=$m: $ms ms/step" | tee -a $O/step_ab.txt
done; done
