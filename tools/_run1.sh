o=gpurun_out/s12; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do
python tools/time_rank_share.py 8 1 > $o/rs81_tf$i.txt 2>&1
CHROMOSIGHT_HIP_PLAN_CONTEND=1 python tools/time_rank_share.py 8 1 > $o/rs81_contend$i.txt 2>&1
python tools/time_rank_share.py 1 0 > $o/rs10_tf$i.txt 2>&1
CHROMOSIGHT_HIP_PLAN_CONTEND=1 python tools/time_rank_share.py 1 0 > $o/rs10_contend$i.txt 2>&1
done
python tools/time_rank_share.py 2 0 > $o/rs20_tf.txt 2>&1
CHROMOSIGHT_HIP_PLAN_CONTEND=1 python tools/time_rank_share.py 2 0 > $o/rs20_contend.txt 2>&1
python tools/time_rank_share.py 4 0 > $o/rs40_tf.txt 2>&1
CHROMOSIGHT_HIP_PLAN_CONTEND=1 python tools/time_rank_share.py 4 0 > $o/rs40_contend.txt 2>&1
CHROMOSIGHT_HIP_TIMING=1 python tools/time_rank_share.py 1 0 2>&1 | grep -E "timing" | tail -20 > $o/genome_host.txt
CHROMOSIGHT_HIP_TIMING=1 python tools/time_rank_share.py 8 1 2>&1 | grep -E "timing" | tail -20 > $o/share_host.txt
tail -n 2 $o/rs*.txt; cat $o/genome_host.txt $o/share_host.txt
