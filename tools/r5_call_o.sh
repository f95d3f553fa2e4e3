#!/bin/bash
# round 5 call O: (1) the attention test file incl. the fused bit-identity test under pytest, (2) does the backward pair get cheaper per
# head when the launch's working set fits the Infinity Cache (chunked-launch hypothesis)?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5o; mkdir -p $O
timeout 900 python -m pytest tests/test_attention_asm_gpu.py -q -x 2>&1 < /dev/null | tail -n 5 | tee $O/pytest_attn.txt
for b in 4 8 16 32 64; do
  timeout 300 python tools/bench_attn_step.py 548 $b 2>&1 < /dev/null | grep "S=" | tee -a $O/bwd_vs_batch.txt
done
