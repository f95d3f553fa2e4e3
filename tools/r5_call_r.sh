#!/bin/bash
# round 5 call R: merged backward launch (MLA_ATTN_BWD_MERGED=<lag>) vs the two-launch form: bits + time
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5r; mkdir -p $O
timeout 300 python tools/exp_attn_bits.py /tmp/bits_two.pt > /dev/null 2>&1 < /dev/null
for lag in 1 2 3; do
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python tools/exp_attn_bits.py /tmp/bits_m$lag.pt > $O/bits_m$lag.log 2>&1 < /dev/null
  echo "== merged lag $lag vs two-launch: $(timeout 120 python tools/exp_attn_bits.py /tmp/bits_two.pt /tmp/bits_m$lag.pt 2>&1 | tail -n 1)" | tee -a $O/merged.txt
done
for rep in 1 2; do
for lag in 0 1 2 3 4; do
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$lag: /" | tee -a $O/merged.txt
done
done
for lag in 0 2; do
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$lag: /" | tee -a $O/merged.txt
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python tools/bench_attn_step.py 548 32 1 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$lag: /" | tee -a $O/merged.txt
done
