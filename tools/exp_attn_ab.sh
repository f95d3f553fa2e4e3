cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in ${SS:-548 2048}; do for tag in ${TAGS:-old new}; do
  lib=$R/mla_amd/libmla_hip.so; [ -d $R/mla_amd/csrc/build_exp/$tag ] && lib=$R/mla_amd/csrc/build_exp/$tag/libmla_hip.so
  rm -rf /tmp/ab_$tag
  MLA_HIP_LIB=$lib MLA_ATTN_BWD5=${FIVE:-0} rocprofv3 --kernel-trace -d /tmp/ab_$tag -o t --output-format rocpd -- python $R/tools/bench_attn_step.py $S 32 > /tmp/ab_$tag.log 2>&1
  echo "== $tag: $(grep 'fwd' /tmp/ab_$tag.log | head -1)"
  python $R/tools/rocpd_stats.py $(find /tmp/ab_$tag -name "*.db" | head -1) | grep -E "attn_" | awk '{printf "   %-70s calls %s avg_us %s\n", $1, $2, $4}'
done; done
