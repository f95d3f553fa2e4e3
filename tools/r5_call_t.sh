#!/bin/bash
# round 5 call T: merged backward launch -- interleaved block order, S = 2048 lag sweep, ragged
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5t; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/tools/exp_attn_bits.py /tmp/bits_two.pt > /dev/null 2>&1 < /dev/null
for m in 5 105; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python $R/tools/exp_attn_bits.py /tmp/bits_m.pt > /dev/null 2>&1 < /dev/null
  echo "== merged $m vs two-launch: $(timeout 120 python $R/tools/exp_attn_bits.py /tmp/bits_two.pt /tmp/bits_m.pt 2>&1 | tail -n 1)" | tee -a $O/sweep.txt
done
for rep in 1 2 3; do
for m in 0 5 105 6 106 103; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python $R/tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$m: /" >> $O/sweep.txt
done
for m in 0 2 4 6 104 106; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python $R/tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$m: /" >> $O/sweep.txt
done
for m in 0 5 105; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python $R/tools/bench_attn_step.py 548 32 1 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$m: /" >> $O/sweep.txt
done
done
grep "S=" $O/sweep.txt | sort | awk '{print $1, $2, $3, $4, $(NF-6), $(NF-5)}'
