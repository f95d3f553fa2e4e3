#!/bin/bash
# A/B of the attention kernels: per-kernel average durations from rocprofv3 kernel traces, old library (build_old/) vs current.
# Usage (GPU box): bash tools/ab_attn.sh [S] [B]
S=${1:-548}; B=${2:-32}
cd /tmp; export TMPDIR=/tmp
for tag in ${TAGS:-old new}; do
  lib=/root/repo/mla_amd/libmla_hip.so; [ -d /root/repo/build_$tag ] && lib=/root/repo/build_$tag/libmla_hip.so
  rm -rf /tmp/ab_$tag
  MLA_HIP_LIB=$lib rocprofv3 --kernel-trace -d /tmp/ab_$tag -o t -- python /root/repo/tools/bench_attn.py $S $B > /tmp/ab_$tag.log 2>&1
  echo "== $tag: $(grep 'fwd' /tmp/ab_$tag.log | head -2 | tr '\n' ' ')"
  python /root/repo/tools/rocpd_stats.py $(find /tmp/ab_$tag -name "*.db" | head -1) | grep -E "attn_" | awk '{printf "   %-60s calls %s avg_us %s\n", $1, $2, $4}'
done
