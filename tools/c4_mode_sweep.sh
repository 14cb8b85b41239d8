#!/bin/bash
# C4 genome step (1 GPU) under the launch-mode switches of the loops pass: default (one launch per block on three streams
# beside the borders chain), one persistent launch for all blocks, other lane counts.   bash tools/c4_mode_sweep.sh <out>
out=${1:-gpurun_out/c4_modes.txt}
: > $out
run() { echo -n "$1: " >> $out; env $2 python bench.py --workload c4 --steps 30 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['patterns'])" >> $out; }
run default X=1

cat $out
