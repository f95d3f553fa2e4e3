# schedule variants of the assembly GEMMs (libraries built with GEN8W_FLAGS / GEN4W_FLAGS=...): correctness, then speed
for d in "" build_8w_spread; do
  echo "== ${d:-default}"
  MLA_HIP_LIB=${d:+$PWD/$d/libmla_hip.so} timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_asm" 2>&1 | tail -1
  MLA_HIP_LIB=${d:+$PWD/$d/libmla_hip.so} python tools/bench_gemm.py 2>&1 | grep -v "qkv fwd  \|TN" | cut -c1-110
done
