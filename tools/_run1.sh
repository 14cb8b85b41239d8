o=gpurun_out/s14; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 500 python tools/fuzz_genomes.py 60 > $o/fuzz_genomes.txt 2>&1
timeout 300 python tools/stress_genome_repeat.py 300 > $o/stress_genome_repeat.txt 2>&1
timeout 300 python tools/stress_pattern_sets.py 40 > $o/stress_pattern_sets.txt 2>&1
timeout 300 python tools/stress_pipeline_threads.py 8 > $o/stress_pipeline_threads.txt 2>&1
timeout 300 python tools/fuzz_detect_options.py 20 7 > $o/fuzz_detect_options.txt 2>&1
timeout 200 python tools/check_quantify_vs_detect.py 3 > $o/check_quantify_vs_detect.txt 2>&1
for f in $o/*.txt; do echo "== $f"; tail -3 $f; done
