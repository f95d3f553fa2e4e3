#!/bin/bash
# round 5, GPU call M: fused one-workgroup-per-head backward (MLA_ATTN_BWD_FUSED=8 / 4) vs the two-kernel form: bits + timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5m; mkdir -p $O
cd $R
timeout 300 python tools/exp_attn_bits.py /tmp/bits_two.pt > $O/bits.txt 2>&1 < /dev/null
for f in 8; do
  MLA_ATTN_BWD_FUSED=$f timeout 300 python tools/exp_attn_bits.py /tmp/bits_f$f.pt >> $O/bits.txt 2>&1 < /dev/null
  echo "== fused $f vs two-kernel: $(timeout 120 python tools/exp_attn_bits.py /tmp/bits_two.pt /tmp/bits_f$f.pt 2>&1 | tail -1)" | tee -a $O/bits.txt
  timeout 120 python tools/exp_attn_cmp.py /tmp/bits_two.pt /tmp/bits_f$f.pt 2>&1 | tail -3 | tee -a $O/bits.txt
done
for r in 1 2; do
  for f in 0 8; do
    MLA_ATTN_BWD_FUSED=$f timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/fused=$f: /" | tee -a $O/timing.txt
  done
done
