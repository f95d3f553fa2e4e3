#!/bin/bash
# AdamW kernel variants (MLA_ADAMW_VARIANT: 0 = product = 4 float4 groups per lane and trip, 9 = one group (rounds 2-3), 1 = two groups,
# 3 / 4 = plain instead of non-temporal accesses with 1 / 2 groups), alternating, two rounds: tools/bench_adamw.py lines (stand-alone, distinct slices, after GEMMs)
for r in 1 2; do for v in 0 9 1 3 4; do echo "== variant $v"; MLA_ADAMW_VARIANT=$v python tools/bench_adamw.py 2>&1 | grep "TB/s"; done; done
