#!/bin/bash
# round 5, GPU call G: LDS swizzle variant of the backward kernels (MLA_ATTN_BWD_SW=1) with counters, + the parity table incl. the generation heads
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5g; mkdir -p $O
X=mla_amd/csrc/build_exp
MLA_HIP_LIB=mla_amd/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_product.pt > $O/bits.txt 2>&1
MLA_HIP_LIB=$X/sw1/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_sw1.pt >> $O/bits.txt 2>&1
echo "== sw1 vs product: $(python tools/exp_attn_bits.py /tmp/bits_product.pt /tmp/bits_sw1.pt 2>&1 | tail -1)" | tee -a $O/bits.txt
for r in 1 2 3; do
  for t in product sw1; do
    lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 548 32 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 8 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  done
done
for t in product sw1; do
  lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=$PWD/mla_amd/libmla_hip.so || lib=$PWD/$lib
  echo "== $t" | tee -a $O/pmc_attn.txt
  MLA_HIP_LIB=$lib GRAFT_REPO_ROOT=$PWD bash tools/pmc_attn_stalls.sh 2>&1 | tee -a $O/pmc_attn.txt
  rm -rf /tmp/pa3
  (cd /tmp && MLA_HIP_LIB=$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pa3 -o p -- python $OLDPWD/tools/bench_attn_step.py > /dev/null 2>&1)
  python - <<PY | tee -a $O/pmc_attn.txt
import csv, collections, glob
f = glob.glob("/tmp/pa3/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    fam = next((k for k in ("attn_fwd", "attn_bwd_dq", "attn_bwd_dkv") if k in r["Kernel_Name"]), None)
    if fam: per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
for fam, d in per.items():
    print(f"  {fam:13s} LDS bank-conflict cycles / LDS index-active cycles = {100 * d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1):.1f} %")
PY
done
python tools/parity_table.py > $O/parity_table.txt 2> $O/parity_table.err; echo "parity rc=$?"; sed -n '/generation heads alone/,$p' $O/parity_table.txt
