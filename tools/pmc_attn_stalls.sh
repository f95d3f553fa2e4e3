# issue-level counters of the three attention kernels as the step calls them (S = 548 by default; ATTN_S=2048 for configs[4])
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pa1 /tmp/pa2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pa1 -o p -- python $R/tools/bench_attn_step.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pa2 -o p -- python $R/tools/bench_attn_step.py > /dev/null 2>&1
python - <<PY
import csv, collections, glob
def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        fam = next((k for k in ("attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_bwd_merged", "attn_delta") if k in n), None)
        if fam is None: continue
        d_ = per[fam][r["Dispatch_Id"]]
        d_[r["Counter_Name"]] = d_.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d_["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return per
p1, p2 = load("/tmp/pa1"), load("/tmp/pa2")
print("attention kernels (all SQ_* in % of SQ_WAVE_CYCLES)")
for fam in p1:
    a = list(p1[fam].values())[1:]; b = list(p2[fam].values())[1:]
    m = lambda ds, k: sum(d.get(k, 0.0) for d in ds) / max(len(ds), 1)
    wc = m(a, "SQ_WAVE_CYCLES"); wc2 = wc
    f = lambda k: 100 * m(a, k) / wc
    print(f"  {fam:13s} {m(a,'_us'):7.1f} us | parked (wait_any) {f('SQ_WAIT_ANY'):5.1f}  issue-stalled (wait_inst_any) {f('SQ_WAIT_INST_ANY'):5.1f} (lds {f('SQ_WAIT_INST_LDS'):4.1f})  issuing {f('SQ_ACTIVE_INST_ANY'):5.1f}"
          f" = valu {f('SQ_ACTIVE_INST_VALU'):5.1f} lds {f('SQ_ACTIVE_INST_LDS'):4.1f} vmem {f('SQ_ACTIVE_INST_VMEM'):4.1f} sca {100*m(b,'SQ_ACTIVE_INST_SCA')/wc:4.1f} misc {100*m(b,'SQ_ACTIVE_INST_MISC')/wc:4.1f}"
          f" | MfmaUtil {100*m(b,'SQ_VALU_MFMA_BUSY_CYCLES')/(m(b,'GRBM_GUI_ACTIVE')/8*1024) if m(b,'GRBM_GUI_ACTIVE') else 0:5.1f}  waves/SIMD {wc/(m(b,'GRBM_GUI_ACTIVE')/8*1024/4) if m(b,'GRBM_GUI_ACTIVE') else 0:4.2f}")
PY
