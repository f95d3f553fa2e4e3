#!/bin/bash
# Whole-step A/B of the opt-in assembly attention forward on BASELINE configs[4] (S = 2048, every layer checkpointed: the forward runs
# twice per layer and step), alternating legs on one box. Output: gpurun_out/r4_config4_attn_asm_ab.txt
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
: > $O/r4_config4_attn_asm_ab.txt
for i in 1 2; do for v in 0 1; do
  MLA_ATTN_FWD=$v timeout 900 python $R/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > /tmp/ab4.json
  python - <<PY >> $O/r4_config4_attn_asm_ab.txt
import json
d = json.load(open("/tmp/ab4.json"))
print("MLA_ATTN_FWD=$v", "ms_per_step", d["ms_per_step"], "samples/s", d["value"], "mfu", d.get("mfu_vs_2.5PF"), "loss", round(d["loss"]["total_loss"], 4))
PY
done; done
cat $O/r4_config4_attn_asm_ab.txt
