# One-rank run of the SHARDED code path (separate shards, RCCL reduce-scatter / all-gather / all-reduce on the side stream):
# A/B against the plain one-process step in the same box + a rocprofv3 step breakdown. Usage (GPU box): bash tools/run_forcecoll.sh <tag>
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
FC="env RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 MLA_FORCE_COLLECTIVES=1"
python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > $O/${TAG}_plain_a.json 2>$O/${TAG}_plain_a.err
$FC python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > $O/${TAG}_forcecoll_a.json 2>$O/${TAG}_forcecoll_a.err
python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > $O/${TAG}_plain_b.json 2>/dev/null
$FC python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > $O/${TAG}_forcecoll_b.json 2>/dev/null
for f in plain_a forcecoll_a plain_b forcecoll_b; do python -c "import json;d=json.load(open('$O/${TAG}_$f.json'));print('$f',d['ms_per_step'],d['rccl_ranks'],d['peak_mem_gb'])"; done
rm -rf /tmp/prof_fc
$FC rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_fc -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > $O/${TAG}_forcecoll_profiled.json 2>/dev/null
python $R/tools/step_breakdown.py $(find /tmp/prof_fc -name "*.db" | head -1) 1 30 > $O/${TAG}_forcecoll_step_breakdown.txt
head -32 $O/${TAG}_forcecoll_step_breakdown.txt
tail -3 $O/${TAG}_forcecoll_a.err
