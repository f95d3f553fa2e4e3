cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in ${SS:-548 2048}; do for five in ${FIVES:-0 1}; do
  rm -rf /tmp/ab_$five
  MLA_ATTN_BWD5=$five rocprofv3 --kernel-trace -d /tmp/ab_$five -o t --output-format rocpd -- python $R/tools/bench_attn_step.py $S 32 > /tmp/ab_$five.log 2>&1
  echo "== five=$five: $(grep 'fwd' /tmp/ab_$five.log | head -1)"
  python $R/tools/rocpd_stats.py $(find /tmp/ab_$five -name "*.db" | head -1) | grep -E "attn_" | awk '{printf "   %-70s calls %s avg_us %s\n", $1, $2, $4}'
done; done
