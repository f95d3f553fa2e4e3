for i in 1 2; do
for lib in mla_amd/csrc/build_tr/libmla_base.so mla_amd/libmla_hip.so; do
  echo "== $lib"; MLA_HIP_LIB=$lib python tools/bench_attn.py 548 32 2>&1 | grep "S="; MLA_HIP_LIB=$lib python tools/bench_attn.py 2048 8 2>&1 | grep "S="
done; done
python -m pytest tests/test_kernels_gpu.py -q -k "attention" 2>&1 | tail -3
