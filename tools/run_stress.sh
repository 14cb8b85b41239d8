#!/bin/bash
# The stress and fuzz tools behind DESIGN.md 2, one after the other (through gpurun from the repo root):
#   bash tools/run_stress.sh <out.txt> [env ...]      e.g.  bash tools/run_stress.sh gpurun_out/stress.txt CHROMOSIGHT_HIP_COUNTS_BAND=1
out=$1; shift
: > $out
run() { echo "== $*" >> $out; env "$@" 2>&1 | tail -${TAILN:-3} >> $out; }
E="$*"
TAILN=3 run $E python tools/fuzz_genomes.py 40
TAILN=2 run $E python tools/stress_genome_repeat.py 150
TAILN=2 run $E python tools/stress_genome_repeat.py 300 30000 300
TAILN=2 run $E python tools/stress_pattern_sets.py 20
TAILN=3 run $E python tools/fuzz_detect_options.py 16
TAILN=1 run $E python tools/stress_two_ranks.py
TAILN=2 run $E python tools/fuzz_edge_inputs.py
TAILN=1 run $E python tools/stress_pipeline_threads.py
cat $out
