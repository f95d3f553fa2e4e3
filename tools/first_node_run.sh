#!/bin/bash
# First hour on the 8 x MI355X node. Nothing in this repo has ever run RCCL with more than one rank: this script
# takes the steps in the order that turns a failure into a diagnosis instead of a hung bench, and ends with ONE table in the shape of
# the driver's SCALE_rNN.json (per-N value, ms/step, scaling efficiency vs N = 1) plus the knob sweep at N = 8.
#
#   bash tools/first_node_run.sh [outdir]          (default gpurun_out/first_node; needs 8 visible devices, ~35-45 min)
#   STEPS=10 WARMUP=3 NS="1 2 4 8" SKIP_TESTS=1 SKIP_KNOBS=1 are honoured.
#
# Order:
#  1. environment check (8 devices, dmabuf IPC switch), library build id
#  2. tests/test_fsdp_2rank_gpu.py::test_two_rccl_ranks_match_single_process -- skipped on every 1-GPU box so far
#  3. tests/test_fsdp_nrank_rccl_gpu.py -- N = 2 / 4 / 8 RCCL ranks vs the same ranks over gloo: drives the nccl branches of
#     mla_amd/fsdp.py (_reduce_scatter: in-place SUM reduce_scatter_tensor; _all_gather: in-place all_gather_into_tensor), compares the
#     reduced shards with gloo's bit for bit (N = 2) / within 4 fp32 ulp, and the out-of-place SUM fallback (MLA_FSDP_INPLACE_RS=0)
#  4. bench.py --gpus 1, 2, 4, 8 (the driver's launch line) with the per-rank wait diagnostics in the JSON line
#  5. the three knobs at N = 8: MLA_RCCL_MAX_CHANNELS (-> NCCL_MAX_NCHANNELS), MLA_FSDP_INPLACE_RS=0, MLA_GEMM_CUS
#  6. the table (tools/first_node_table.py)
# Reference: training/strategies/fsdp.py:88-93 (sharding strategy names), :181-209 (FSDP wrapping), :308-310 (clip).
set -u
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/first_node}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
STEPS=${STEPS:-10}; WARMUP=${WARMUP:-3}; NS=${NS:-"1 2 4 8"}
log() { echo "[$(date +%H:%M:%S)] $*" | tee -a "$OUT/log.txt"; }

# ---- 1. environment
NDEV=$(python -c 'import torch; print(torch.cuda.device_count())')
log "devices visible: $NDEV; HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY"
python - <<PY 2>&1 | tee -a "$OUT/log.txt"
import sys; sys.path.insert(0, "$R")
from mla_amd import hip
print("libmla_hip.so arch", hip.lib().mla_query(1), "gemm source id", hip.gemm_source_id())
PY
rocm-smi --showtopo > "$OUT/topology.txt" 2>&1 || true
if [ "$NDEV" -lt 2 ]; then log "fewer than 2 devices: nothing to do here"; exit 2; fi

# ---- 2. + 3. correctness of the RCCL branches before any timing
if [ -z "${SKIP_TESTS:-}" ]; then
  log "2. two RCCL ranks vs one process"
  (cd "$R" && timeout 1500 python -m pytest tests/test_fsdp_2rank_gpu.py -q -x -s -k rccl) > "$OUT/test_2rank_rccl.txt" 2>&1
  log "   rc=$? ($(tail -1 "$OUT/test_2rank_rccl.txt"))"
  log "3. N = 2 / 4 / 8 RCCL ranks vs gloo ranks (in-place SUM reduce-scatter, in-place all-gather, out-of-place fallback)"
  (cd "$R" && timeout 3000 python -m pytest tests/test_fsdp_nrank_rccl_gpu.py -q -s) > "$OUT/test_nrank_rccl.txt" 2>&1
  rc=$?
  log "   rc=$rc ($(tail -1 "$OUT/test_nrank_rccl.txt"))"
  grep -h "^RCCL x" "$OUT/test_nrank_rccl.txt" | tee -a "$OUT/log.txt"
  if [ $rc -ne 0 ]; then
    log "   RCCL path differs from gloo: re-running the bench sweep below with MLA_FSDP_INPLACE_RS=0 as well; look at test_nrank_rccl.txt first"
  fi
fi

# ---- 4. the scaling sweep, exactly as the driver launches it
bench() {   # bench <tag> <N> [ENV=VAL ...]
  local tag=$1 n=$2; shift 2
  local port=$((29600 + RANDOM % 300))
  if [ "$n" = 1 ]; then
    env "$@" timeout 1500 python "$R/bench.py" --gpus 1 --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-secondary > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  else
    env "$@" timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      "$R/bench.py" --gpus $n --steps $STEPS --warmup $WARMUP > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  fi
  local rc=$?
  log "   bench $tag (N=$n $*): rc=$rc $(python -c "import json,sys; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['value'], d['unit'], (d.get('per_rank') or {}).get('rs_event_wait_ms_per_step'))" 2>/dev/null)"
  [ $rc -ne 0 ] && tail -5 "$OUT/bench_$tag.err" | tee -a "$OUT/log.txt"
}
log "4. bench.py --gpus $NS ($STEPS timed steps, $WARMUP warm-up)"
for n in $NS; do
  [ "$n" -le "$NDEV" ] && bench "n$n" $n
done

# ---- 5. knobs at the largest N
NMAX=$(for n in $NS; do [ "$n" -le "$NDEV" ] && echo $n; done | tail -1)
if [ -z "${SKIP_KNOBS:-}" ] && [ "$NMAX" -gt 1 ]; then
  log "5. knob sweep at N = $NMAX"
  for ch in 8 16 32; do bench "n${NMAX}_chan$ch" $NMAX MLA_RCCL_MAX_CHANNELS=$ch; done
  bench "n${NMAX}_avg_rs" $NMAX MLA_FSDP_INPLACE_RS=0
  for cus in 240 224; do bench "n${NMAX}_cus$cus" $NMAX MLA_GEMM_CUS=$cus; done
fi

# ---- 6. one table
python "$R/tools/first_node_table.py" "$OUT" | tee "$OUT/SCALE_table.txt"
log "done: $OUT/SCALE_table.txt, $OUT/SCALE_first_node.json"
