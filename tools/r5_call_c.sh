#!/bin/bash
# round 5, GPU call C: attention backward A/B -- base (round-4 kernels) vs the round-5 prologue / epilogue / pipelined-fragment variants
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c; mkdir -p $O
X=mla_amd/csrc/build_exp
# 1. bit identity of every variant against the round-4 library
MLA_HIP_LIB=$X/base/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_base.pt > $O/bits.txt 2>&1
for t in product pf0 kv4 kv8 kv10; do
  lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
  MLA_HIP_LIB=$lib python tools/exp_attn_bits.py /tmp/bits_$t.pt >> $O/bits.txt 2>&1
  echo "== $t vs base: $(python tools/exp_attn_bits.py /tmp/bits_base.pt /tmp/bits_$t.pt 2>&1 | tail -1)" | tee -a $O/bits.txt
done
# 2. timing, alternating, the form the step calls
for r in 1 2 3; do
  for t in base product pf0 kv4 kv8 kv10; do
    lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 548 32 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  done
done
for t in base product pf0 kv8; do
  lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
  MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 8 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 32 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
done
# 3. block-phase trace of the new kernels
MLA_HIP_LIB=$X/btrace2/libmla_hip.so python tools/exp_attn_btrace.py 548 32 > $O/btrace_548_new.txt 2>&1
grep -v amdgpu.ids $O/btrace_548_new.txt
# 4. attention tests on the product library
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention or attn" > $O/test_attn.txt 2>&1; tail -n 3 $O/test_attn.txt
