#!/bin/bash
# Slowest-rank step time of the C4 genome at 1 / 2 / 4 / 8 shares, each share timed alone on ONE GPU (no exchange):
#   bash tools/rank_share_sweep.sh <out.txt>
out=${1:-gpurun_out/rank_share.txt}
: > $out
for w in 1 2 4 8; do
  for r in $(seq 0 $((w-1))); do
    python tools/time_rank_share.py $w $r 2>/dev/null | grep "GPUs, rank" | tail -1 >> $out
  done
done
python - "$out" <<'PY'
import re, sys
worst, worst_med = {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"(\d+) GPUs, rank (\d+): .* step ([\d.]+) ms \(median ([\d.]+)", line)
    if m:
        w, t, md = int(m.group(1)), float(m.group(3)), float(m.group(4))
        worst[w] = max(worst.get(w, 0.0), t)
        worst_med[w] = max(worst_med.get(w, 0.0), md)
with open(sys.argv[1], "a") as f:
    # (a share is timed over 20 steps: one step that the host lost to another process moves a mean by 10 % -- the medians say
    # what a rank does, the means what this run saw)
    for w in sorted(worst):
        f.write(f"slowest share at {w}: {worst[w]:.3f} ms  ({worst[1] / worst[w]:.2f}x of the 1-share step); "
                f"by the medians {worst_med[w]:.3f} ms ({worst_med[1] / worst_med[w]:.2f}x)\n")
PY
tail -4 $out
