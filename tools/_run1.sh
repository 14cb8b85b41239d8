o=gpurun_out/s13; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $o/test.log
python tools/time_c5_phases.py > $o/c5_phases.txt 2>&1
python tools/prof_c5.py > $o/prof_c5.txt 2>&1
python bench.py --workload c5 --steps 20 --warmup 3 2>/dev/null | tail -1 > $o/bench_c5.json
cat $o/test.log $o/c5_phases.txt; head -40 $o/prof_c5.txt | cut -c1-150
