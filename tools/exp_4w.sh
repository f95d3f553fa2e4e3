# schedule variants of the 4-wave assembly GEMM (libraries built with GEN4W_FLAGS=...): correctness (bit-identical to gemm256), then speed
for d in "" build_4w_loadsfirst build_4w_midbarrier; do
  echo "== ${d:-spread}"
  MLA_HIP_LIB=${d:+$PWD/$d/libmla_hip.so} timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_asm and 5" 2>&1 | tail -1
  MLA_HIP_LIB=${d:+$PWD/$d/libmla_hip.so} python tools/bench_gemm.py 2>&1 | grep -E "o wgrad|qkv wgrad|down dgrad|gu fwd" | cut -c1-100
done
