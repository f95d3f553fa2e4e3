cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "Warning\|warnings.warn\|^$" > gpurun_out/gputest1.log; tail -5 gpurun_out/gputest1.log
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.err; tail -c 600 gpurun_out/bench_c1.json
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --keep-layers 0 > gpurun_out/bench_c4_k0.json 2> gpurun_out/bench_c4_k0.err; python -c "import json;d=json.load(open('gpurun_out/bench_c4_k0.json'));print('c4 k0',d['ms_per_step'],d['peak_mem_gb'])"
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/bench_c4_auto.json 2> gpurun_out/bench_c4_auto.err; python -c "import json;d=json.load(open('gpurun_out/bench_c4_auto.json'));print('c4 auto',d['ms_per_step'],d['peak_mem_gb'],d['config'].get('activation_policy'))"
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --keep-level 1 > gpurun_out/bench_c4_l1.json 2> gpurun_out/bench_c4_l1.err; python -c "import json;d=json.load(open('gpurun_out/bench_c4_l1.json'));print('c4 lvl1',d['ms_per_step'],d['peak_mem_gb'],d['config'].get('activation_policy'))"
tail -3 gpurun_out/*.err | tail -20
