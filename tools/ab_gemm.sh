# A/B of two builds of the library in one box (power/clock differences between boxes exceed the effects measured)
for rep in 1 2; do
echo "== new"; python tools/bench_gemm.py 2>&1 | grep -v "qkv fwd  \|TN" | cut -c1-75
echo "== old"; MLA_HIP_LIB=$PWD/build_old/libmla_hip.so python tools/bench_gemm.py 2>&1 | grep -v "qkv fwd  \|TN" | cut -c1-75
done
