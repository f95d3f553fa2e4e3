#!/bin/bash
# round 5, GPU call K: register staging of the Q / dO tiles in the dK dV kernel vs LDS-DMA staging (same source otherwise)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5k; mkdir -p $O
cd $R
X=mla_amd/csrc/build_exp
MLA_HIP_LIB=$X/rs0/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_rs0.pt > $O/bits.txt 2>&1 < /dev/null
MLA_HIP_LIB=mla_amd/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_product.pt >> $O/bits.txt 2>&1 < /dev/null
echo "== product (register staging) vs rs0 (LDS-DMA): $(python tools/exp_attn_bits.py /tmp/bits_rs0.pt /tmp/bits_product.pt 2>&1 | tail -1)" | tee -a $O/bits.txt
for r in 1 2 3; do
  for t in rs0 product; do
    lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  done
done
MLA_HIP_LIB=$X/btrace2/libmla_hip.so python tools/exp_attn_btrace.py 548 32 > $O/btrace_548.txt 2>&1 < /dev/null
grep -v amdgpu.ids $O/btrace_548.txt | sed -n '/dK.dV/,$p'
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention or attn" > $O/test_attn.txt 2>&1 < /dev/null; tail -n 2 $O/test_attn.txt
