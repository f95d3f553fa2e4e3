# gemm256 vs hipBLASLt (torch.matmul) on the same shapes under the same counters: where does the library kernel's edge come from?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ -n "$ONE" ]; then set -- "$ONE"; else set -- "16384 4096 4096" "17536 4096 4096" "4096 11008 17536" "17536 12288 4096"; fi
for shape in "$@"; do
  rm -rf /tmp/pg1 /tmp/pg2
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pg1 -o p -- python $R/tools/pmc_gemm.py $shape > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pg2 -o p -- python $R/tools/pmc_gemm.py $shape > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pg3 -o p -- python $R/tools/pmc_gemm.py $shape > /dev/null 2>&1
  echo "== M N K = $shape"
  python - <<PY
import csv, collections, re, glob
def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        fam = "gemm256" if "gemm256_kernel" in n else "asm8w" if "gemm_asm_kernel" in n or "gemm8w" in n else "asm4w" if "gemm4w" in n else ("hipBLASLt " + re.sub(r"^.*?(MT\d+x\d+x\d+).*$", r"\1", n)[:20] if "Cijk" in n else None)
        if fam is None: continue
        d_ = per[fam][r["Dispatch_Id"]]
        v = float(r["Counter_Value"]); s, m = d_.get(r["Counter_Name"], (0.0, 0.0)); d_[r["Counter_Name"]] = (s + v, max(m, v))
        d_["_us"] = ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,) * 2
    return per
p1, p2, p3 = load("/tmp/pg1"), load("/tmp/pg2"), load("/tmp/pg3")
M, N, K = map(int, "$shape".split())
for fam in p1:
    ds = list(p1[fam].values())[1:]      # drop the first (cold) launch
    mean = lambda k, i=0: sum(d[k][i] for d in ds) / len(ds)
    us = mean("_us"); gui = mean("GRBM_GUI_ACTIVE", 1)
    if gui / (us * 1e3) > 4: gui = mean("GRBM_GUI_ACTIVE") / 8
    hit = sum(d["TCC_HIT_sum"][0] for d in list(p2[fam].values())[1:]); miss = sum(d["TCC_MISS_sum"][0] for d in list(p2[fam].values())[1:])
    f3 = list(p3[fam].values())[1:]; rd = sum(d["FETCH_SIZE"][0] for d in f3) / len(f3) * 2048
    print(f"  {fam:32s} {us:8.1f} us  {2.0*M*N*K/us/1e9:7.1f} TF/s(pmc)  clk {gui/(us*1e3):.2f} GHz  MfmaUtil {100*mean('SQ_VALU_MFMA_BUSY_CYCLES')/(gui*1024):5.1f}%  "
          f"VALU {100*mean('SQ_ACTIVE_INST_VALU')/(256*gui):5.1f}%  wait {100*mean('SQ_WAIT_ANY')/mean('SQ_WAVE_CYCLES'):5.1f}%  LDSconf {100*mean('SQ_LDS_BANK_CONFLICT')/max(mean('SQ_LDS_IDX_ACTIVE'),1):4.1f}%  "
          f"LDSidx/flop {mean('SQ_LDS_IDX_ACTIVE')/(2.0*M*N*K)*1e3:.3f}  L2hit {100*hit/(hit+miss):5.1f}%  fabric rd {rd/1e9:.2f} GB  waves {mean('SQ_WAVE_CYCLES')/gui/256:.1f}/CU")
PY
done
