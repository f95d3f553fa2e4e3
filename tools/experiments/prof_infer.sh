# rocprofv3 kernel stats of tools/bench_infer.py (GPU box): bash tools/experiments/prof_infer.sh <tag>
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_inf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o p -- python $GRAFT_REPO_ROOT/tools/bench_infer.py --iters 3 > /tmp/inf.log 2>&1
f=$(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/r6_infer_kernel_stats_$TAG.csv
python - <<EOF2
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r6_infer_kernel_stats_$TAG.csv")))
for r in rows[:12]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), "avg us", round(float(r["AverageNs"])/1e3,1), "total ms", round(float(r["TotalDurationNs"])/1e6,1), r["Percentage"])
EOF2
tail -1 /tmp/inf.log | cut -c1-400
