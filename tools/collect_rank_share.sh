#!/bin/bash
# Rank-share evidence of the C4 genome on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_rank_share.sh <out dir under gpurun_out>
# one share of 8 and the whole genome: step times, device timelines (rocprofv3 kernel trace), native lap times, C4 / C5 phases
o=${1:-gpurun_out/rs}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/$o"
cd /tmp && export TMPDIR=/tmp
cd "$root"
python tools/time_rank_share.py 8 1 > $o/rs81.txt 2>&1
python tools/time_rank_share.py 1 0 > $o/rs10.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/kt_8_1 -o rs -- python tools/time_rank_share.py 8 1 > $o/kt_8_1.log 2>&1
python tools/kernel_timeline.py $o/kt_8_1 > $o/rank_share_timeline.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/kt_1_0 -o rs -- python tools/time_rank_share.py 1 0 > $o/kt_1_0.log 2>&1
python tools/kernel_timeline.py $o/kt_1_0 > $o/genome_timeline.txt 2>&1
CHROMOSIGHT_HIP_TIMING=1 python tools/time_rank_share.py 8 1 2>&1 | grep timing | tail -24 > $o/rank_share_native_laps.txt
python tools/time_c4_phases.py 6 > $o/c4_phases.txt 2>&1
python tools/time_c5_phases.py > $o/c5_phases.txt 2>&1
rm -rf $o/kt_8_1 $o/kt_1_0
cat $o/rs81.txt $o/rs10.txt
