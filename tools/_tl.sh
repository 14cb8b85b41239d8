cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for wr in "1 0" "8 5"; do
  set -- $wr
  rm -rf /tmp/kt_$1
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$1 -o rs -- python tools/time_rank_share.py $1 $2 > /tmp/kt_$1.log 2>&1
  python tools/kernel_timeline.py /tmp/kt_$1 > gpurun_out/timeline_final_$1.txt 2>&1
done
cut -c1-110 gpurun_out/timeline_final_1.txt | head -30
