# in-step kernel times (AdamW, plain GEMM, RMSNorm backward) against how ShardedModel places its buffers (MLA_FSDP_ARENAS)
cd /tmp; export TMPDIR=/tmp
for e in "MLA_FSDP_ARENAS=0" "MLA_FSDP_ARENAS=master,exp_avg,exp_avg_sq,grad32,flat16" "MLA_FSDP_ARENAS=master,exp_avg,exp_avg_sq"; do
  rm -rf /tmp/pa; env $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > /tmp/pa.json 2>/dev/null
  echo "== $e: $(python -c "import json;print(json.load(open('/tmp/pa.json'))['ms_per_step'])") ms/step"
  python - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob('/tmp/pa/**/*kernel_stats.csv', recursive=True)[0])):
    if any(k in r["Name"] for k in ("adamw_vec4", "gemm256_kernel<0, 0, 0, true>", "rmsnorm_bwd_kernel<2>")):
        print(f'   {r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
done
