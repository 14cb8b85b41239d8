o=gpurun_out/s20; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CHROMOSIGHT_HIP_NO_STEP_PLAN=1
python tools/time_rank_share.py 1 0 > $o/interp_default.txt 2>&1
CHROMOSIGHT_HIP_NARROW_STAGING=1 python tools/time_rank_share.py 1 0 > $o/interp_narrow.txt 2>&1
CHROMOSIGHT_HIP_NARROW_STAGING=1 CHROMOSIGHT_HIP_BLOCK_TABLE=1 python tools/time_rank_share.py 1 0 > $o/interp_narrow_table.txt 2>&1
CHROMOSIGHT_HIP_BLOCK_TABLE=1 python tools/time_rank_share.py 1 0 > $o/interp_table.txt 2>&1
export CHROMOSIGHT_HIP_NARROW_STAGING=1 CHROMOSIGHT_HIP_BLOCK_TABLE=1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/kt -o rs -- python tools/time_rank_share.py 1 0 > $o/kt.log 2>&1
python tools/kernel_timeline.py $o/kt > $o/timeline_narrow_table.txt 2>&1
rm -rf $o/kt
for f in $o/interp*.txt; do echo "$(basename $f): $(tail -1 $f | grep -o 'step.*')"; done
cat $o/timeline_narrow_table.txt | cut -c1-110
