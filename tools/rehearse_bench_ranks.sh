#!/bin/bash
# Rehearsal of bench.py's N-rank control flow on ONE GPU (first-run insurance for the 8-GPU node): the exact launch
# line the driver uses (python -m torch.distributed.run ... bench.py --gpus N), with MLA_BENCH_REHEARSAL=1 = every rank on device 0, gloo
# instead of RCCL, tiny model. Exercises: rank bookkeeping, barriers, FSDP reduce-scatter / all-gather per unit, per-rank diagnostics,
# max-over-ranks timing, exactly one JSON line on stdout from rank 0. Usage (on the GPU box): bash tools/rehearse_bench_ranks.sh [N ...]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in ${@:-2 8}; do
  port=$((29500 + n))
  MLA_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
    --master-port $port $R/bench.py --gpus $n --steps 3 --warmup 1 --tiny --no-cpu-baseline --no-secondary > /tmp/rehearse_$n.out 2> /tmp/rehearse_$n.err
  echo "== N=$n rc=$? stdout lines: $(wc -l < /tmp/rehearse_$n.out)"
  python - <<PY
import json
line = open("/tmp/rehearse_$n.out").read().strip().splitlines()
assert len(line) == 1, line
d = json.loads(line[0])
print({k: d.get(k) for k in ("n_gpus", "rccl_ranks", "ms_per_step", "value", "rehearsal")})
print("per_rank:", d.get("per_rank")); print("knobs:", d.get("collective_knobs")); print("loss:", d.get("loss"))
PY
  tail -3 /tmp/rehearse_$n.err
done
