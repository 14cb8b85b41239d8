#!/bin/bash
# Kernel-only timings of the benched correlation calls (C2, C3k, C4', C3 from CSR) on the GPU box:
#   gpurun -- 'bash tools/quick_kernels.sh <tag>'   -> gpurun_out/quick_<tag>.txt
tag=${1:-run}
out=gpurun_out/quick_${tag}.txt
mkdir -p gpurun_out
: > $out
for w in c2 c3k c4p c3; do
  python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>>gpurun_out/quick_${tag}.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$w', 'ms', d['kernel_ms'], 'frac', r['frac'], 'kernel', r['kernel_id'])" >> $out
done
cat $out
