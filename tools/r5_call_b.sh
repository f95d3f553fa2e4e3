#!/bin/bash
# round 5, GPU call B: block-phase traces of the attention backward kernels + the re-bounded parity tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5b; mkdir -p $O
MLA_HIP_LIB=mla_amd/csrc/build_exp/btrace/libmla_hip.so python tools/exp_attn_btrace.py 548 32 > $O/btrace_548.txt 2>&1; echo "btrace548 rc=$?"
MLA_HIP_LIB=mla_amd/csrc/build_exp/btrace/libmla_hip.so python tools/exp_attn_btrace.py 2048 8 > $O/btrace_2048.txt 2>&1; echo "btrace2048 rc=$?"
timeout 900 python -m pytest tests/test_generation_gpu.py -q -s -k "7b_dimensions or post_training" > $O/test_gen.txt 2>&1; echo "gen rc=$?"
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "reference_golden or cropped" > $O/test_model.txt 2>&1; echo "model rc=$?"
cat $O/btrace_548.txt; tail -5 $O/test_gen.txt $O/test_model.txt
