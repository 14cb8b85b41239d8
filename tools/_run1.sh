cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do
for r in 1 6; do
echo "== block rank $r: $(python tools/time_rank_share.py 8 $r 2>/dev/null | tail -1 | grep -o 'step.*;')"
echo "== spin  rank $r: $(CHROMOSIGHT_HIP_SPIN_WAIT=1 python tools/time_rank_share.py 8 $r 2>/dev/null | tail -1 | grep -o 'step.*;')"
done
done
echo "== block genome: $(python tools/time_rank_share.py 1 0 2>/dev/null | tail -1 | grep -o 'step.*;')"
echo "== spin  genome: $(CHROMOSIGHT_HIP_SPIN_WAIT=1 python tools/time_rank_share.py 1 0 2>/dev/null | tail -1 | grep -o 'step.*;')"
CHROMOSIGHT_HIP_SPIN_WAIT=1 CHROMOSIGHT_HIP_TIMING=1 python tools/time_rank_share.py 8 1 2>&1 | grep -E "wait|enqueued|lane 0 call  [78]" | tail -5
