# fused d(act) + SwiGLU-backward epilogue: where do its ~390 us per launch go? (experiment build: MLA_EXPERIMENTAL=1 bash mla_amd/csrc/build.sh mla_amd/csrc/build_exp)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MLA_HIP_LIB=$GRAFT_REPO_ROOT/mla_amd/csrc/build_exp/libmla_hip.so
for a in 0 16 8 1 2 3 4 7 15 0; do echo -n "MLA_EPI_ABL=$a  "; MLA_EPI_ABL=$a python tools/exp_stagger.py 2>/dev/null | tail -1; done
