#!/bin/bash
# sample clocks/power while the GEMM micro-benchmark runs
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > gpurun_out/clk_samples.jsonl &
SP=$!
python tools/bench_gemm.py > gpurun_out/gemm_clk.log 2>&1
kill $SP 2>/dev/null
tail -14 gpurun_out/gemm_clk.log
python - <<'PY'
import json
for line in open('gpurun_out/clk_samples.jsonl'):
    try: d=json.loads(line)
    except Exception: continue
    c=d.get('card0',{})
    print({k:v for k,v in c.items() if 'sclk' in k.lower() or 'power' in k.lower() or 'mclk' in k.lower()})
PY
