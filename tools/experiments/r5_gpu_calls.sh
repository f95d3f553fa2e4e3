#!/bin/bash
# The 25 one-shot GPU lease scripts of round 5 (tools/r5_call_a.sh .. z.sh) folded into ONE parametrised script:
#   bash tools/experiments/r5_gpu_calls.sh <letter>        e.g.  gpurun -- 'bash tools/experiments/r5_gpu_calls.sh m'
# Each case is the record of what one round-5 GPU call ran (HISTORY.md "Round 5" cites them by letter). Round 6 uses tools/ab_pairs.py
# (alternating same-box A/B pairs with a confidence interval) and tools/collect_counters.sh instead of new one-shot scripts.
call=${1:?usage: r5_gpu_calls.sh <letter a..z>}
case "$call" in
a)
# round 5, GPU call A: parity table, the new parity tests, the default bench line (with the secondary block), attention baseline
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5a; mkdir -p $O
python tools/parity_table.py > $O/parity_table.txt 2> $O/parity_table.err; echo "parity rc=$?"
timeout 900 python -m pytest tests/test_generation_gpu.py -q -x -s -k "7b_dimensions" > $O/test_imggen7b.txt 2>&1; echo "imggen7b rc=$?"
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "cropped" > $O/test_crop.txt 2>&1; echo "crop rc=$?"
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python tools/bench_attn.py 548 32 > $O/attn.txt 2>&1; python tools/bench_attn.py 2048 8 >> $O/attn.txt 2>&1
tail -3 $O/test_imggen7b.txt $O/test_crop.txt; cat $O/attn.txt; cut -c1-400 $O/bench_default.json
;;
b)
# round 5, GPU call B: block-phase traces of the attention backward kernels + the re-bounded parity tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5b; mkdir -p $O
MLA_HIP_LIB=mla_amd/csrc/build_exp/btrace/libmla_hip.so python tools/exp_attn_btrace.py 548 32 > $O/btrace_548.txt 2>&1; echo "btrace548 rc=$?"
MLA_HIP_LIB=mla_amd/csrc/build_exp/btrace/libmla_hip.so python tools/exp_attn_btrace.py 2048 8 > $O/btrace_2048.txt 2>&1; echo "btrace2048 rc=$?"
timeout 900 python -m pytest tests/test_generation_gpu.py -q -s -k "7b_dimensions or post_training" > $O/test_gen.txt 2>&1; echo "gen rc=$?"
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "reference_golden or cropped" > $O/test_model.txt 2>&1; echo "model rc=$?"
cat $O/btrace_548.txt; tail -5 $O/test_gen.txt $O/test_model.txt
;;
c)
# round 5, GPU call C: attention backward A/B -- base (round-4 kernels) vs the round-5 prologue / epilogue / pipelined-fragment variants
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c; mkdir -p $O
X=mla_amd/csrc/build_exp
# 1. bit identity of every variant against the round-4 library
MLA_HIP_LIB=$X/base/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_base.pt > $O/bits.txt 2>&1
for t in product pf0 kv4 kv8 kv10; do
  lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
  MLA_HIP_LIB=$lib python tools/exp_attn_bits.py /tmp/bits_$t.pt >> $O/bits.txt 2>&1
  echo "== $t vs base: $(python tools/exp_attn_bits.py /tmp/bits_base.pt /tmp/bits_$t.pt 2>&1 | tail -1)" | tee -a $O/bits.txt
done
# 2. timing, alternating, the form the step calls
for r in 1 2 3; do
  for t in base product pf0 kv4 kv8 kv10; do
    lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 548 32 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  done
done
for t in base product pf0 kv8; do
  lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
  MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 8 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 32 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
done
# 3. block-phase trace of the new kernels
MLA_HIP_LIB=$X/btrace2/libmla_hip.so python tools/exp_attn_btrace.py 548 32 > $O/btrace_548_new.txt 2>&1
grep -v amdgpu.ids $O/btrace_548_new.txt
# 4. attention tests on the product library
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention or attn" > $O/test_attn.txt 2>&1; tail -n 3 $O/test_attn.txt
;;
d)
# round 5, GPU call D: attention backward v2 (rows through the LDS-DMA path, o^T from the prologue) vs the round-4 kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5d; mkdir -p $O
X=mla_amd/csrc/build_exp
MLA_HIP_LIB=$X/base/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_base.pt > $O/bits.txt 2>&1
MLA_HIP_LIB=mla_amd/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_product.pt >> $O/bits.txt 2>&1
echo "== product vs base: $(python tools/exp_attn_bits.py /tmp/bits_base.pt /tmp/bits_product.pt 2>&1 | tail -1)" | tee -a $O/bits.txt
for r in 1 2 3; do
  for t in base product; do
    lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 548 32 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 548 32 1 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  done
done
for t in base product; do
  lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
  MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 8 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 32 2>&1 | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
done
MLA_HIP_LIB=$X/btrace2/libmla_hip.so python tools/exp_attn_btrace.py 548 32 > $O/btrace_548_new.txt 2>&1
grep -v amdgpu.ids $O/btrace_548_new.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention or attn" > $O/test_attn.txt 2>&1; tail -n 3 $O/test_attn.txt
;;
e)
# round 5, GPU call E: the whole -m gpu suite on the v2 attention kernels + the default bench line + same-box A/B of the step against the round-4 library
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5e; mkdir -p $O
bash tools/run_gpu_tests.sh > $O/gputest_summary.txt 2>&1; cp gpurun_out/gputest.log $O/gputest.log
cat $O/gputest_summary.txt | head -20
for r in 1 2; do
  MLA_HIP_LIB=mla_amd/csrc/build_exp/base/libmla_hip.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_base_$r.json 2>> $O/bench.err
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_new_$r.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do echo "$f $(python -c "import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_mfu'] if 'whole_step_mfu' in d['roofline'] else '')")"; done
MLA_HIP_LIB=mla_amd/csrc/build_exp/base/libmla_hip.so python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench4_base.json 2>> $O/bench.err
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench4_new.json 2>> $O/bench.err
MLA_ATTN_FWD=0 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench4_new_fwd0.json 2>> $O/bench.err
for f in $O/bench4_*.json; do echo "$f $(python -c "import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"; done
;;
g)
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
;;
h)
# round 5, GPU call H: final build -- whole -m gpu suite, lga_prep_bwd timing, counters / profiles of record (tag r5)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
bash tools/run_gpu_tests.sh > $O/r5_gputest_summary.txt 2>&1; cp $O/gputest.log $O/r5_gpu_test_log.txt; head -3 $O/r5_gputest_summary.txt
python tools/bench_lga_prep_bwd.py > $O/r5_lga_prep_bwd_timing.txt 2>&1; cat $O/r5_lga_prep_bwd_timing.txt | grep stage
bash tools/collect_counters.sh r5 all > $O/r5_collect.log 2>&1; tail -6 $O/r5_collect.log; cat $O/r5_pmc_table.txt | tail -10
;;
i)
# round 5, GPU call I: in-step per-kernel attention times, swizzle 0 vs swizzle 1 builds, same box, alternating (rocprofv3 kernel stats of 4 steps)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5i; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for r in 1 2; do
  for t in sw0 sw1p; do
    lib=$R/mla_amd/csrc/build_exp/sw0/libmla_hip.so; [ $t = sw1p ] && lib=$R/mla_amd/libmla_hip.so
    rm -rf /tmp/pi_$t$r
    MLA_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi_$t$r -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > $O/bench_$t$r.json 2>/dev/null < /dev/null
    f=$(find /tmp/pi_$t$r -name "*kernel_stats.csv" 2>/dev/null | head -1)
    echo "== $t run $r: $(python -c "import json;print(json.loads(open('$O/bench_$t$r.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>&1 | tail -1) ms/step" | tee -a $O/attn_instep.txt
    [ -n "$f" ] && python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'attn_' in r['Name'] and 'local' not in r['Name']: print('   %-40s calls %s avg %.1f us' % (r['Name'].split('::')[1][:40], r['Calls'], float(r['AverageNs'])/1e3))
" | tee -a $O/attn_instep.txt
  done
done
;;
j)
# round 5, GPU call J: N-rank rehearsal of bench.py after this round's edits + the new true-dimension point-cloud head test
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_generation_gpu.py -q -s -k "true_dimensions or 7b_dimensions or heads_against" > $O/test_gen.txt 2>&1 < /dev/null; echo "gen rc=$?"; tail -n 4 $O/test_gen.txt
timeout 1200 bash tools/rehearse_bench_ranks.sh 2 8 > $O/rehearsal.txt 2>&1 < /dev/null; echo "rehearsal rc=$?"; grep -v "^\[W\|Warning\|warn" $O/rehearsal.txt | tail -n 16
;;
k)
# round 5, GPU call K: register staging of the Q / dO tiles in the dK dV kernel vs LDS-DMA staging (same source otherwise)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5k; mkdir -p $O
cd $R
X=mla_amd/csrc/build_exp
MLA_HIP_LIB=$X/rs0/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_rs0.pt > $O/bits.txt 2>&1 < /dev/null
MLA_HIP_LIB=mla_amd/libmla_hip.so python tools/exp_attn_bits.py /tmp/bits_product.pt >> $O/bits.txt 2>&1 < /dev/null
echo "== product (register staging) vs rs0 (LDS-DMA): $(python tools/exp_attn_bits.py /tmp/bits_rs0.pt /tmp/bits_product.pt 2>&1 | tail -1)" | tee -a $O/bits.txt
for r in 1 2 3; do
  for t in rs0 product; do
    lib=$X/$t/libmla_hip.so; [ $t = product ] && lib=mla_amd/libmla_hip.so
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
    MLA_HIP_LIB=$lib python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/$t: /" | tee -a $O/timing.txt
  done
done
MLA_HIP_LIB=$X/btrace2/libmla_hip.so python tools/exp_attn_btrace.py 548 32 > $O/btrace_548.txt 2>&1 < /dev/null
grep -v amdgpu.ids $O/btrace_548.txt | sed -n '/dK.dV/,$p'
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention or attn" > $O/test_attn.txt 2>&1 < /dev/null; tail -n 2 $O/test_attn.txt
;;
l)
# round 5, GPU call L: one block per CU (one wave per SIMD) -- per-tile cost of a wave that has the CU to itself
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5l; mkdir -p $O
cd $R
for e in 0 1; do
  echo "== MLA_ATTN_BWD_LDS_EXTRA=$e" | tee -a $O/one_block.txt
  MLA_ATTN_BWD_LDS_EXTRA=$e python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | tee -a $O/one_block.txt
  MLA_ATTN_BWD_LDS_EXTRA=$e python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | tee -a $O/one_block.txt
  MLA_ATTN_BWD_LDS_EXTRA=$e MLA_HIP_LIB=mla_amd/csrc/build_exp/btrace2/libmla_hip.so python tools/exp_attn_btrace.py 2048 8 2>&1 < /dev/null | grep -v amdgpu | grep "==\|  all\|     16 \|     32 \|CUs seen" | tee -a $O/one_block.txt
done
;;
m)
# round 5, GPU call M: fused one-workgroup-per-head backward (MLA_ATTN_BWD_FUSED=8 / 4) vs the two-kernel form: bits + timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5m; mkdir -p $O
cd $R
timeout 300 python tools/exp_attn_bits.py /tmp/bits_two.pt > $O/bits.txt 2>&1 < /dev/null
for f in 8; do
  MLA_ATTN_BWD_FUSED=$f timeout 300 python tools/exp_attn_bits.py /tmp/bits_f$f.pt >> $O/bits.txt 2>&1 < /dev/null
  echo "== fused $f vs two-kernel: $(timeout 120 python tools/exp_attn_bits.py /tmp/bits_two.pt /tmp/bits_f$f.pt 2>&1 | tail -1)" | tee -a $O/bits.txt
  timeout 120 python tools/exp_attn_cmp.py /tmp/bits_two.pt /tmp/bits_f$f.pt 2>&1 | tail -3 | tee -a $O/bits.txt
done
for r in 1 2; do
  for f in 0 8; do
    MLA_ATTN_BWD_FUSED=$f timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/fused=$f: /" | tee -a $O/timing.txt
  done
done
;;
n)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5n; mkdir -p $O
for f in 8; do
MLA_ATTN_BWD_FUSED=$f MLA_HIP_LIB=$R/mla_amd/csrc/build_exp/btrace2/libmla_hip.so timeout 300 python tools/exp_attn_fused_trace.py 2>&1 < /dev/null | grep -v amdgpu | tee -a $O/fused_trace.txt
MLA_ATTN_BWD_FUSED=$f timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/fused=$f: /" | tee -a $O/fused_trace.txt
done
timeout 300 python tools/exp_attn_bits.py /tmp/bits_two.pt > /dev/null 2>&1 < /dev/null
MLA_ATTN_BWD_FUSED=8 timeout 300 python tools/exp_attn_bits.py /tmp/bits_f8.pt > /dev/null 2>&1 < /dev/null
echo "== fused 8 vs two-kernel: $(timeout 120 python tools/exp_attn_bits.py /tmp/bits_two.pt /tmp/bits_f8.pt 2>&1 | tail -1)" | tee -a $O/fused_trace.txt
;;
o)
# round 5 call O: (1) the attention test file incl. the fused bit-identity test under pytest, (2) does the backward pair get cheaper per
# head when the launch's working set fits the Infinity Cache (chunked-launch hypothesis)?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5o; mkdir -p $O
timeout 900 python -m pytest tests/test_attention_asm_gpu.py -q -x 2>&1 < /dev/null | tail -n 5 | tee $O/pytest_attn.txt
for b in 4 8 16 32 64; do
  timeout 300 python tools/bench_attn_step.py 548 $b 2>&1 < /dev/null | grep "S=" | tee -a $O/bwd_vs_batch.txt
done
;;
p)
# round 5 call P: the round-end sequence on the final commit -- whole GPU suite, smoke, the default bench line (with `secondary`),
# then the 120-step soak.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5p2; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 < /dev/null | tail -n 15 > $O/gpu_test_log.txt; tail -n 3 $O/gpu_test_log.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | tail -n 3 | tee $O/smoke.txt
( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err < /dev/null; tail -n 4 $O/bench_default.err; cut -c1-400 $O/bench_default.json
timeout 900 python tools/soak.py 120 2>&1 < /dev/null | grep -v amdgpu > $O/soak_120steps.txt; tail -n 3 $O/soak_120steps.txt
;;
q)
# round 5 call Q: front-end timeline of one traced configs[1] step (what runs before the first decoder-layer attention)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5q; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_q
timeout 900 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_q -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile > $O/bench.json 2> $O/bench.err < /dev/null
DB=$(find /tmp/prof_q -name "*.db" | head -1)
if [ -n "$DB" ]; then timeout 300 python $R/tools/step_frontend_timeline.py $DB 1 > $O/frontend_timeline.txt 2>&1; fi
tail -n 5 $O/frontend_timeline.txt; cut -c1-200 $O/bench.json
;;
r)
# round 5 call R: merged backward launch (MLA_ATTN_BWD_MERGED=<lag>) vs the two-launch form: bits + time
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5r; mkdir -p $O
timeout 300 python tools/exp_attn_bits.py /tmp/bits_two.pt > /dev/null 2>&1 < /dev/null
for lag in 1 2 3; do
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python tools/exp_attn_bits.py /tmp/bits_m$lag.pt > $O/bits_m$lag.log 2>&1 < /dev/null
  echo "== merged lag $lag vs two-launch: $(timeout 120 python tools/exp_attn_bits.py /tmp/bits_two.pt /tmp/bits_m$lag.pt 2>&1 | tail -n 1)" | tee -a $O/merged.txt
done
for rep in 1 2; do
for lag in 0 1 2 3 4; do
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$lag: /" | tee -a $O/merged.txt
done
done
for lag in 0 2; do
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$lag: /" | tee -a $O/merged.txt
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python tools/bench_attn_step.py 548 32 1 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$lag: /" | tee -a $O/merged.txt
done
;;
s)
# round 5 call S: merged backward launch -- lag sweep (3 repetitions) and fabric bytes (FETCH_SIZE / WRITE_SIZE) merged vs two launches
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5s; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in 1 2 3; do
for lag in 0 2 3 4 5 6 8; do
  MLA_ATTN_BWD_MERGED=$lag timeout 300 python $R/tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$lag: /" >> $O/lag_sweep.txt
done
done
cat $O/lag_sweep.txt | sort | awk '{print $1, $(NF-6), $(NF-5)}' | tail -n 30
for m in 0 3; do
 for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$m$c
  MLA_ATTN_BWD_MERGED=$m timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$m$c -o p -- python $R/tools/bench_attn_step.py 548 32 > /dev/null 2>&1 < /dev/null
  echo "== merged=$m $c" >> $O/pmc_bytes.txt
  python $R/tools/pmc_summary.py $(find /tmp/pm_$m$c -name "*counter_collection.csv" | head -1) 'attn_(fwd|bwd)_\w+kernel' >> $O/pmc_bytes.txt 2>&1
 done
done
cat $O/pmc_bytes.txt
;;
t)
# round 5 call T: merged backward launch -- interleaved block order, S = 2048 lag sweep, ragged
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5t; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/tools/exp_attn_bits.py /tmp/bits_two.pt > /dev/null 2>&1 < /dev/null
for m in 5 105; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python $R/tools/exp_attn_bits.py /tmp/bits_m.pt > /dev/null 2>&1 < /dev/null
  echo "== merged $m vs two-launch: $(timeout 120 python $R/tools/exp_attn_bits.py /tmp/bits_two.pt /tmp/bits_m.pt 2>&1 | tail -n 1)" | tee -a $O/sweep.txt
done
for rep in 1 2 3; do
for m in 0 5 105 6 106 103; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python $R/tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$m: /" >> $O/sweep.txt
done
for m in 0 2 4 6 104 106; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python $R/tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$m: /" >> $O/sweep.txt
done
for m in 0 5 105; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python $R/tools/bench_attn_step.py 548 32 1 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$m: /" >> $O/sweep.txt
done
done
grep "S=" $O/sweep.txt | sort | awk '{print $1, $2, $3, $4, $(NF-6), $(NF-5)}'
;;
u)
# round 5 call U: whole-step same-box A/B of the merged backward launch: configs[1] (3 alternating rounds) and configs[4]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5u; mkdir -p $O
for r in 1 2 3; do for m in 0 105; do
  ms=$(MLA_ATTN_BWD_MERGED=$m timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-gemm-profile 2>/dev/null < /dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "configs[1] MLA_ATTN_BWD_MERGED=$m: $ms ms/step" | tee -a $O/step_ab.txt
done; done
for r in 1 2; do for m in 0 104; do
  ms=$(MLA_ATTN_BWD_MERGED=$m timeout 900 python bench.py --config 4 --keep-layers 0 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-gemm-profile 2>/dev/null < /dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "configs[4] MLA_ATTN_BWD_MERGED=$m: $ms ms/step" | tee -a $O/step_ab.txt
done; done
;;
v)
# round 5 call V: merged backward launch as the default -- whole GPU suite, smoke, stand-alone timing, default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5v; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 < /dev/null | tail -n 15 > $O/gpu_test_log.txt; tail -n 3 $O/gpu_test_log.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | tail -n 2 | tee $O/smoke.txt
for m in 0 105 0 105; do
  MLA_ATTN_BWD_MERGED=$m timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/merged=$m: /" | tee -a $O/timing.txt
done
( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err < /dev/null; tail -n 4 $O/bench_default.err; cut -c1-300 $O/bench_default.json
;;
w)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5w; mkdir -p $O
timeout 3000 python -X faulthandler -m pytest tests/ -v -m gpu > $O/gpu_test_full.txt 2>&1 < /dev/null
grep -c "PASSED" $O/gpu_test_full.txt; grep -n "FAILED\|ERROR" $O/gpu_test_full.txt | head -n 10; tail -n 3 $O/gpu_test_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | tail -n 1
;;
x)
# round 5 call X: profiles of record on the final build (merged backward launch): kernel stats + step breakdown + roofline table + PMC
# table + traffic records + configs 3 / 4 (tools/collect_counters.sh, tools/collect_profiles.sh), tag r5f
R=${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=$R
timeout 1500 bash $R/tools/collect_counters.sh r5f > /dev/null 2>&1
timeout 2400 bash $R/tools/collect_profiles.sh r5f > /dev/null 2>&1
ls -la $R/gpurun_out | grep r5f_ | awk '{print $5, $9}'
head -n 12 $R/gpurun_out/r5f_step_breakdown_config1.txt | cut -c1-120
grep -i "attn" $R/gpurun_out/r5f_pmc_table.txt | cut -c1-250
;;
y)
# round 5 call Y: timing probe -- merged launch with the head counter bumped at block start (results wrong, timing valid): the upper bound
# of what an earlier delta buys at small lags
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5y; mkdir -p $O
for rep in 1 2 3; do
for m in 105 101 102 103 1 2; do
  MLA_HIP_LIB=$R/mla_amd/csrc/build_exp/pubearly/libmla_hip.so MLA_ATTN_BWD_MERGED=$m timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/early m=$m: /" >> $O/probe.txt
done
MLA_ATTN_BWD_MERGED=105 timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/product m=105: /" >> $O/probe.txt
done
for m in 104 101 102; do
  MLA_HIP_LIB=$R/mla_amd/csrc/build_exp/pubearly/libmla_hip.so MLA_ATTN_BWD_MERGED=$m timeout 300 python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/early m=$m: /" >> $O/probe.txt
done
sort $O/probe.txt | awk '{print $1, $2, $3, $4, $(NF-5)}'
;;
z)
# round 5 call Z: static wave priority in the merged backward launch (1: dQ blocks, 2: dK.dV blocks, 3: every other block per XCD)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5z; mkdir -p $O
for rep in 1 2 3; do
for v in product prio1 prio2 prio3; do
  lib=$R/mla_amd/csrc/build_exp/$v/libmla_hip.so; [ $v = product ] && lib=$R/mla_amd/libmla_hip.so
  MLA_HIP_LIB=$lib timeout 300 python tools/bench_attn_step.py 548 32 2>&1 < /dev/null | grep "S=" | sed "s/^/$v: /" >> $O/prio.txt
  MLA_HIP_LIB=$lib timeout 300 python tools/bench_attn_step.py 2048 8 2>&1 < /dev/null | grep "S=" | sed "s/^/$v: /" >> $O/prio.txt
done
done
sort $O/prio.txt | awk '{print $1, $2, $3, $(NF-5)}'
;;
*) echo "unknown call '$call'"; exit 2 ;;
esac
