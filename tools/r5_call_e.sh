#!/bin/bash
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
