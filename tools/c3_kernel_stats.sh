#!/bin/bash
# Per-kernel averages of the C3-from-CSR step (rocprofv3 --kernel-trace --stats), with the band of counts and without:
#   gpurun -- 'bash tools/c3_kernel_stats.sh <tag>'   -> gpurun_out/c3_stats_<tag>.txt
tag=${1:-run}
root=$(pwd)
out=$root/gpurun_out/c3_stats_$tag.txt
cd /tmp && export TMPDIR=/tmp
cd "$root"
: > $out
for mode in counts detrended; do
  envs=""; [ "$mode" = detrended ] && envs="CS_BENCH_C3_DETRENDED=1"
  rm -rf /tmp/c3prof_$mode
  env $envs timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3prof_$mode -o c3 -- \
      python bench.py --workload c3 --steps 50 --warmup 10 --no-cpu-baseline > /tmp/c3prof_$mode.log 2>&1
  echo "== $mode" >> $out
  f=$(find /tmp/c3prof_$mode -name "*kernel_stats.csv" | head -1)
  python - "$f" >> $out <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
  tail -1 /tmp/c3prof_$mode.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], d['kernel_ms'])" >> $out
done
cat $out
