# A/B of the GEMM dispatch in one box (power/clock differences between boxes exceed the effects measured)
for rep in 1 2; do
echo "== persistent"; MLA_GEMM_PERSIST=1 python tools/bench_gemm.py 2>&1 | grep -v "qkv fwd  \|TN" | cut -c1-75
echo "== one workgroup per tile"; python tools/bench_gemm.py 2>&1 | grep -v "qkv fwd  \|TN" | cut -c1-75
done
