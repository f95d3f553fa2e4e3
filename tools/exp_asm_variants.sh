cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in ${TAGS:-exp hbl exp hbl}; do
  export MLA_HIP_LIB=$GRAFT_REPO_ROOT/mla_amd/csrc/build_exp/$t/libmla_hip.so
  [ -n "$TESTS" ] && python -m pytest tests/test_kernels_gpu.py -q -k "gemm_asm_kernel" 2>&1 | tail -1
  python tools/exp_asm_variants.py 2>/dev/null | tail -1
done
