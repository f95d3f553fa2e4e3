# issue-level counters of one GEMM shape for gemm256 / asm kernels / hipBLASLt: where do a wave's cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${ONE:-"4096 4096 17536"}
rm -rf /tmp/ps1 /tmp/ps2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/ps1 -o p -- python $R/tools/pmc_gemm.py $SHAPE > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/ps2 -o p -- python $R/tools/pmc_gemm.py $SHAPE > /dev/null 2>&1
python - <<PY
import csv, collections, re, glob
def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        fam = "gemm256" if "gemm256_kernel" in n else "asm8w" if "gemm8w" in n else "asm4w" if "gemm4w" in n else ("hipBLASLt" if "Cijk" in n else None)
        if fam is None: continue
        d_ = per[fam][r["Dispatch_Id"]]
        d_[r["Counter_Name"]] = d_.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d_["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return per
p1, p2 = load("/tmp/ps1"), load("/tmp/ps2")
print("shape $SHAPE  (all SQ_* in % of SQ_WAVE_CYCLES: quad-cycles summed over waves)")
for fam in p1:
    a = list(p1[fam].values())[1:]; b = list(p2[fam].values())[1:]
    m = lambda ds, k: sum(d.get(k, 0.0) for d in ds) / len(ds)
    wc = m(a, "SQ_WAVE_CYCLES")
    f = lambda k: 100 * m(a, k) / wc
    print(f"  {fam:10s} {m(a,'_us'):7.1f} us | wait_any {f('SQ_WAIT_ANY'):5.1f}  wait_inst_any {f('SQ_WAIT_INST_ANY'):5.1f} (lds {f('SQ_WAIT_INST_LDS'):4.1f})  active_any {f('SQ_ACTIVE_INST_ANY'):5.1f}"
          f"  valu {f('SQ_ACTIVE_INST_VALU'):5.1f}  lds {f('SQ_ACTIVE_INST_LDS'):4.1f}  vmem {f('SQ_ACTIVE_INST_VMEM'):4.1f} | sca {100*m(b,'SQ_ACTIVE_INST_SCA')/wc:4.1f} misc {100*m(b,'SQ_ACTIVE_INST_MISC')/wc:4.1f}"
          f"  vmem_rd_cyc {100*m(b,'SQ_INST_CYCLES_VMEM_RD')/wc:5.1f}  ifetch {m(b,'SQ_IFETCH'):.3g}  lvl_vmem {m(b,'SQ_INST_LEVEL_VMEM')/wc:.2f} lvl_lds {m(b,'SQ_INST_LEVEL_LDS')/wc:.2f}")
PY
