cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in 0:0 1:2000 1:4500 1:7000 2:2000 2:4500 2:7000 0:0 1:4500 2:4500; do MLA_GEMM_STAGGER=$s python tools/exp_stagger.py 2>/dev/null | tail -1; done
