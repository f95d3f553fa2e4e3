#!/bin/bash
# round 5, GPU call L: one block per CU (one wave per SIMD) -- per-tile cost of a wave that has the CU to itself
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5l; mkdir -p $O
cd $R
for e in 0 1; do
  echo "== MLA_ATTN_BWD_LDS_EXTRA=$e" | tee -a $O/one_block.txt
  MLA_ATTN_BWD_LDS_EXTRA=$e python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | tee -a $O/one_block.txt
  MLA_ATTN_BWD_LDS_EXTRA=$e python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | tee -a $O/one_block.txt
  MLA_ATTN_BWD_LDS_EXTRA=$e MLA_HIP_LIB=mla_amd/csrc/build_exp/btrace2/libmla_hip.so python tools/exp_attn_btrace.py 2048 8 2>&1 < /dev/null | grep -v amdgpu | grep "==\|  all\|     16 \|     32 \|CUs seen" | tee -a $O/one_block.txt
done
