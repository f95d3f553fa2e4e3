# Round 6: the one-launch attention backward under CU contention (VERDICT r5 next #1c). Runs tools/contention_rehearsal.py --quick with
# the merged launch (default) and with MLA_ATTN_BWD_MERGED=0 in the same box; the per-case hex losses / gradient norms of the two runs
# must be identical (same kernels' block bodies, deterministic step) and no run may trap. Usage (GPU box): bash tools/contention_merged.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
python $R/tools/contention_rehearsal.py --quick --steps 5 > $O/r6_contention_merged_run.txt 2> $O/r6_contention_merged_run.err; echo "merged rc=$?"
MLA_ATTN_BWD_MERGED=0 python $R/tools/contention_rehearsal.py --quick --steps 5 > $O/r6_contention_two_run.txt 2> $O/r6_contention_two_run.err; echo "two-launch rc=$?"
{
  echo "== one launch (default) =="; grep -v '^{' $O/r6_contention_merged_run.txt
  echo; echo "== two launches (MLA_ATTN_BWD_MERGED=0), same box =="; grep -v '^{' $O/r6_contention_two_run.txt
  echo; echo "== per-case results (hex loss / grad norm of the last step), merged vs two launches =="
  if diff <(grep 'last step' $O/r6_contention_merged_run.txt) <(grep 'last step' $O/r6_contention_two_run.txt) > /dev/null; then echo "IDENTICAL in all $(grep -c 'last step' $O/r6_contention_merged_run.txt) cases"; else echo "DIFFERENT"; diff <(grep 'last step' $O/r6_contention_merged_run.txt) <(grep 'last step' $O/r6_contention_two_run.txt); fi
} > $O/r6_contention_merged.txt
tail -5 $O/r6_contention_merged_run.err
cat $O/r6_contention_merged.txt
