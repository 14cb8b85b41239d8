o=gpurun_out/s11; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/time_rank_share.py 8 1 > $o/rs81_default.txt 2>&1
CHROMOSIGHT_HIP_PLAN_TILES_FIRST=1 python tools/time_rank_share.py 8 1 > $o/rs81_tf.txt 2>&1
python tools/time_rank_share.py 1 0 > $o/rs10_default.txt 2>&1
CHROMOSIGHT_HIP_PLAN_TILES_FIRST=1 python tools/time_rank_share.py 1 0 > $o/rs10_tf.txt 2>&1
CHROMOSIGHT_HIP_PLAN_TILES_FIRST=1 CHROMOSIGHT_HIP_BLOCK_TABLE=1 python tools/time_rank_share.py 1 0 > $o/rs10_tf_table.txt 2>&1
CHROMOSIGHT_HIP_BLOCK_TABLE=1 python tools/time_rank_share.py 1 0 > $o/rs10_table.txt 2>&1
python tools/time_rank_share.py 8 1 > $o/rs81_default2.txt 2>&1
CHROMOSIGHT_HIP_PLAN_TILES_FIRST=1 python tools/time_rank_share.py 8 1 > $o/rs81_tf2.txt 2>&1
export CHROMOSIGHT_HIP_PLAN_TILES_FIRST=1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/kt_8_1 -o rs -- python tools/time_rank_share.py 8 1 > $o/kt_8_1.log 2>&1
python tools/kernel_timeline.py $o/kt_8_1 > $o/rank_share_timeline_tf.txt 2>&1
export CHROMOSIGHT_HIP_BLOCK_TABLE=1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/kt_1_0 -o rs -- python tools/time_rank_share.py 1 0 > $o/kt_1_0.log 2>&1
python tools/kernel_timeline.py $o/kt_1_0 > $o/genome_timeline_tf_table.txt 2>&1
rm -rf $o/kt_8_1 $o/kt_1_0
unset CHROMOSIGHT_HIP_PLAN_TILES_FIRST CHROMOSIGHT_HIP_BLOCK_TABLE
python tools/prof_c5.py > $o/prof_c5.txt 2>&1
tail -n 2 $o/rs*.txt; cat $o/rank_share_timeline_tf.txt
