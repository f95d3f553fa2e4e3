cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/exp_stagger.py 2>/dev/null | tail -1
python -m pytest tests/test_kernels_gpu.py -q -k "swiglu or acts" 2>&1 | tail -2
python tools/exp_stagger.py 2>/dev/null | tail -1
