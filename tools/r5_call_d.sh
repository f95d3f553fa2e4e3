#!/bin/bash
# round 5, GPU call D: attention backward v2 (rows through the LDS-DMA path, o^T from the prologue) vs the round-4 kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5d; mkdir -p $O
X=mla_amd/csrc/build_exp
MLA_HIP_LIB=$X/base/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_base.pt > $O/bits.txt 2>&1
MLA_HIP_LIB=mla_amd/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_product.pt >> $O/bits.txt 2>&1
echo "== product vs base: $(python tools/exp_attn_bits.py /tmp/bits_base.pt /tmp/bits_product.pt 2>&1 | tail -1)" | tee -a $O/bits.txt
for r in 1 2 3; do
  for t in base product; do
    lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 548 32 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 548 32 1 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  done
done
for t in base product; do
  lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
  MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 8 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 32 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
done
MLA_HIP_LIB=$X/btrace2/libmla_hip.so python tools/exp_attn_btrace.py 548 32 > $O/btrace_548_new.txt 2>&1
grep -v amdgpu.ids $O/btrace_548_new.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention or attn" > $O/test_attn.txt 2>&1; tail -n 3 $O/test_attn.txt
