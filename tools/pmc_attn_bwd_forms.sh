#!/bin/bash
# Counters of the three forms of the attention backward at the step's shape (S = 548, B' = 32, 32 heads x 128; RoPE backward and the four
# transposed wgrad operands written by the kernels), experiment build: two launches (dQ, dK.dV), the merged launch (product default), and the
# one-workgroup-per-head kernel that reads q, k, v, dO, o once (MLA_ATTN_BWD_FUSED=8; round 5). One rocprofv3 pass per counter set
# (--pmc is never combined with tracing domains other than --kernel-trace). Usage (GPU box): bash tools/pmc_attn_bwd_forms.sh > out.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MLA_HIP_LIB=$R/mla_amd/csrc/explib/libmla_hip.so
for form in two merged fused; do
  case $form in two) E="MLA_ATTN_BWD_MERGED=0";; merged) E="MLA_ATTN_BWD_MERGED=105";; fused) E="MLA_ATTN_BWD_FUSED=8";; esac
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); rm -rf /tmp/pf_${form}_$i
    env $E rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pf_${form}_$i -o p -- python $R/tools/bench_attn_step.py > /tmp/pf_${form}_$i.log 2>&1
  done
  env $E python $R/tools/bench_attn_step.py 2>&1 | grep "S=" | head -1 | sed "s/^/[$form, un-profiled] /"
done
python - <<'PY'
import csv, glob, collections
def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    if not f: return per
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "attn_bwd" not in n and "attn_delta" not in n: continue
        import re
        key = (re.findall(r"attn_[a-z0-9_]+", n) or [n[:40]])[0]
        d_ = per[key][r["Dispatch_Id"]]
        d_[r["Counter_Name"]] = d_.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d_["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return per
print("attention backward, S = 548, B' = 32, 1 024 heads; per LAUNCH, mean over the launches after the first (separate --pmc passes)")
print(f"{'form / kernel':58s} {'us':>8s} {'fabric rd GB':>12s} {'fabric wr GB':>12s} {'MfmaUtil %':>10s} {'wait_any %':>10s} {'LDS confl %':>11s} {'L2 hit %':>8s}")
for form in ("two", "merged", "fused"):
    P = [load(f"/tmp/pf_{form}_{i}") for i in (1, 2, 3, 4)]
    tot = dict(us=0.0, rd=0.0, wr=0.0)
    for k in P[2]:
        m = lambda p, c: (lambda ds: sum(d.get(c, 0.0) for d in ds) / max(len(ds), 1))(list(p[k].values())[1:] or list(p[k].values()))
        us = m(P[2], "_us"); rd = m(P[0], "FETCH_SIZE") * 1024 * 2 / 1e9; wr = m(P[1], "WRITE_SIZE") * 1024 / 1e9
        gui = m(P[2], "GRBM_GUI_ACTIVE"); wc = m(P[2], "SQ_WAVE_CYCLES")
        mf = 100 * m(P[2], "SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024) if gui else float("nan")
        hit, mis = m(P[3], "TCC_HIT_sum"), m(P[3], "TCC_MISS_sum")
        print(f"{form + ' / ' + k:58s} {us:8.1f} {rd:12.3f} {wr:12.3f} {mf:10.1f} {100 * m(P[2], 'SQ_WAIT_ANY') / wc if wc else 0:10.1f} "
              f"{100 * m(P[2], 'SQ_LDS_BANK_CONFLICT') / max(m(P[2], 'SQ_LDS_IDX_ACTIVE'), 1):11.1f} {100 * hit / max(hit + mis, 1):8.1f}")
        tot["us"] += us; tot["rd"] += rd; tot["wr"] += wr
    print(f"{form + ' / TOTAL':58s} {tot['us']:8.1f} {tot['rd']:12.3f} {tot['wr']:12.3f}   (single-read minimum: 0.72 GB read, 1.01 GB written)")
PY
