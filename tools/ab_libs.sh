#!/bin/bash
# A/B of library builds on the benched kernels, interleaved, three repetitions:
#   bash tools/ab_libs.sh <out.txt> "<workloads>" <lib-or-'-'> <lib2> ...      ('-' = the in-tree library)
out=$1; wls=$2; shift 2
: > $out
for rep in 1 2 3; do
  for lib in "$@"; do
    for w in $wls; do
      envs=""; [ "$lib" != "-" ] && envs="CHROMOSIGHT_HIP_LIBRARY=$lib"
      env $envs python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$lib] $w:', d['kernel_ms'], r['frac'])" >> $out
    done
  done
done
cat $out
