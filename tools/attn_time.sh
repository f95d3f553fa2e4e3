cd /tmp && export TMPDIR=/tmp
for t in ${TS:-1}; do rm -rf /tmp/p$t; MLA_ATTN_BWD_T=$t rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$t -o x -- python /root/repo/bench.py --steps 4 --warmup 1 > /tmp/p$t.log 2>&1; echo "== T=$t"; f=$(find /tmp/p$t -name "*kernel_stats.csv" | head -1); grep -E "attn_" $f | cut -d, -f1-4 | cut -c1-150; done
