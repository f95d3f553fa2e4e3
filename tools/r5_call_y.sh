#!/bin/bash
# round 5 call Y: timing probe -- merged launch with the head counter bumped at block start (results wrong, timing valid): the upper bound
# of what an earlier delta buys at small lags
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5y; mkdir -p $O
for rep in 1 2 3; do
for m in 105 101 102 103 1 2; do
  MLA_HIP_LIB=$R/mla_amd/csrc/build_exp/pubearly/libmla_hip.so MLA_ATTN_BWD_MERGED=$m timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/early m=$m: /" >> $O/probe.txt
done
MLA_ATTN_BWD_MERGED=105 timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/product m=105: /" >> $O/probe.txt
done
for m in 104 101 102; do
  MLA_HIP_LIB=$R/mla_amd/csrc/build_exp/pubearly/libmla_hip.so MLA_ATTN_BWD_MERGED=$m timeout 300 python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/early m=$m: /" >> $O/probe.txt
done
sort $O/probe.txt | awk '{print $1, $2, $3, $4, $(NF-5)}'
