#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5n; mkdir -p $O
for f in 8; do
MLA_ATTN_BWD_FUSED=$f MLA_HIP_LIB=$R/mla_amd/csrc/build_exp/btrace2/libmla_hip.so timeout 300 python tools/exp_attn_fused_trace.py 2>&1 < /dev/null | grep -v amdgpu | tee -a $O/fused_trace.txt
MLA_ATTN_BWD_FUSED=$f timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/fused=$f: /" | tee -a $O/fused_trace.txt
done
timeout 300 python tools/exp_attn_bits.py /tmp/bits_two.pt > /dev/null 2>&1 < /dev/null
MLA_ATTN_BWD_FUSED=8 timeout 300 python tools/exp_attn_bits.py /tmp/bits_f8.pt > /dev/null 2>&1 < /dev/null
echo "== fused 8 vs two-kernel: $(timeout 120 python tools/exp_attn_bits.py /tmp/bits_two.pt /tmp/bits_f8.pt 2>&1 | tail -1)" | tee -a $O/fused_trace.txt
