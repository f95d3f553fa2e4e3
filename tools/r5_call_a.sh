#!/bin/bash
# round 5, GPU call A: parity table, the new parity tests, the default bench line (with the secondary block), attention baseline
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5a; mkdir -p $O
python tools/parity_table.py > $O/parity_table.txt 2> $O/parity_table.err; echo "parity rc=$?"
timeout 900 python -m pytest tests/test_generation_gpu.py -q -x -s -k "7b_dimensions" > $O/test_imggen7b.txt 2>&1; echo "imggen7b rc=$?"
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "cropped" > $O/test_crop.txt 2>&1; echo "crop rc=$?"
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python tools/bench_attn.py 548 32 > $O/attn.txt 2>&1; python tools/bench_attn.py 2048 8 >> $O/attn.txt 2>&1
tail -3 $O/test_imggen7b.txt $O/test_crop.txt; cat $O/attn.txt; cut -c1-400 $O/bench_default.json
