#!/bin/bash
# round 5 call Z: static wave priority in the merged backward launch (1: dQ blocks, 2: dK.dV blocks, 3: every other block per XCD)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5z; mkdir -p $O
for rep in 1 2 3; do
for v in product prio1 prio2 prio3; do
  lib=$R/mla_amd/csrc/build_exp/$v/libmla_hip.so; [ $v = product ] && lib=$R/mla_amd/libmla_hip.so
  MLA_HIP_LIB=$lib timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/$v: /" >> $O/prio.txt
  MLA_HIP_LIB=$lib timeout 300 python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/$v: /" >> $O/prio.txt
done
done
sort $O/prio.txt | awk '{print $1, $2, $3, $(NF-5)}'
