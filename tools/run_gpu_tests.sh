# full GPU suite (no -x: report every failure), output filtered into gpurun_out/gputest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s "$@" 2>&1 | grep -v "Warning\|warnings.warn\|amdgpu.ids\|^$\|socket.cpp\|Gloo\] Rank" > gpurun_out/gputest.log
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/gputest.log | tail -30
grep -n "reduced gradients\|reduce-scatter(mean)\|2 ranks (\|decoder layer @7B\|clip norm" gpurun_out/gputest.log | cut -c1-900
