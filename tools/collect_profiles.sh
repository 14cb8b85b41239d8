#!/bin/bash
# Round measurement collection on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02
# Writes raw profiler output under gpurun_out/<tag>/ ; tools/summarize_profiles.py copies the
# summaries into profiles/ (tracked).  Counter passes are separate from the trace passes and from
# each other (FETCH_SIZE and WRITE_SIZE do not fit one pass), as the MI355X guide prescribes.
set -u
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"

bench() { timeout 600 python bench.py "$@" 2>/dev/null | tail -1; }

# 1. bench lines (the default command first: it is what the driver runs)
bench > "$out/bench_c2.json"
bench --workload c3 --no-cpu-baseline > "$out/bench_c3.json"
bench --workload c3k --no-cpu-baseline > "$out/bench_c3k.json"
bench --workload c4p --no-cpu-baseline > "$out/bench_c4p.json"
bench --workload c4 --steps 40 --warmup 8 > "$out/bench_c4.json"
bench --workload c5 --steps 10 --warmup 2 > "$out/bench_c5.json"
bench --size 16384 --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_dense16384.json"
bench --precision f64 --no-cpu-baseline > "$out/bench_c2_f64.json"
CHROMOSIGHT_HIP_NO_MFMA=1 CHROMOSIGHT_HIP_NO_SYMMETRY=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_c2_nosym.json"
# the packed-FMA streaming kernel on the workloads whose default is a matrix-core kernel (dense map; banded maps with
# the mirrored 17 x 17 loops template)
CHROMOSIGHT_HIP_NO_MFMA=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_c2_stream.json"
CHROMOSIGHT_HIP_MFMA_REG=0 python bench.py --workload c3k --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_c3k_stream.json"
CHROMOSIGHT_HIP_MFMA_REG=0 python bench.py --workload c4p --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_c4p_stream.json"

# 2. kernel traces of the same commands (c3 and c4 show the helper kernels: band extents, distance law,
#    tiler, mask tables, compaction, sort, foci, re-scoring)
for w in c2 c3 c3k c4p; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -o $w -- \
      python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > "$out/trace_$w.log" 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c4" -o c4 -- \
    python bench.py --workload c4 --steps 3 --warmup 1 > "$out/trace_c4.log" 2>&1

# 3. HBM counters of the dominant kernel of the dense and the two banded workloads, one pass each
for w in c2 c3k c4p; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_${w}_$c" -o $w -- \
        python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > "$out/pmc_${w}_$c.log" 2>&1
  done
  # VALU / wave counters (own pass)
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY \
      --kernel-trace --output-format csv -d "$out/pmc_${w}_sq" -o $w -- \
      python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > "$out/pmc_${w}_sq.log" 2>&1
done
# matrix-core occupancy of the dense kernel (own pass)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --kernel-trace --output-format csv -d "$out/pmc_c2_mfma" -o c2 -- \
    python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline > "$out/pmc_c2_mfma.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --kernel-trace --output-format csv -d "$out/pmc_c4p_mfma" -o c4p -- \
    python bench.py --workload c4p --steps 5 --warmup 2 --no-cpu-baseline > "$out/pmc_c4p_mfma.log" 2>&1
# the candidate (CAND) instances the pipeline runs -- corr_mfma_dense_kernel<.., CAND> per block, corr_mfma_blocks_kernel for
# a rank's share -- on the C4 genome: matrix-core occupancy and HBM bytes (own passes)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d "$out/pmc_c4_mfma" -o c4 -- \
    python bench.py --workload c4 --steps 3 --warmup 1 > "$out/pmc_c4_mfma.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_c4_$c" -o c4 -- \
      python bench.py --workload c4 --steps 3 --warmup 1 > "$out/pmc_c4_$c.log" 2>&1
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d "$out/pmc_share_mfma" -o share -- \
    python tools/time_rank_share.py 8 1 > "$out/pmc_share_mfma.log" 2>&1
# rank shares of the C4 genome, each alone on this GPU; host + device timeline of one share of 8; C4 / C5 phases; the
# per-phase cycle shares of the tile kernels (profiling build)
bash tools/rank_share_sweep.sh "$out/rank_share.txt" > /dev/null 2>&1
python tools/trace_rank_share.py 8 1 > "$out/rank_share_host_trace.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out/kt_8_1" -o rs -- python tools/time_rank_share.py 8 1 > "$out/kt_8_1.log" 2>&1
python tools/kernel_timeline.py "$out/kt_8_1" > "$out/rank_share_timeline.txt" 2>&1
CHROMOSIGHT_HIP_TIMING=1 python tools/time_rank_share.py 8 1 2>&1 | grep timing | tail -19 > "$out/rank_share_native_laps.txt"
# the whole genome on one GPU: device timeline of one step, host timeline of its call list
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out/kt_1_0" -o rs -- python tools/time_rank_share.py 1 0 > "$out/kt_1_0.log" 2>&1
python tools/kernel_timeline.py "$out/kt_1_0" > "$out/genome_timeline.txt" 2>&1
CHROMOSIGHT_HIP_TIMING=1 python tools/time_rank_share.py 1 0 2>&1 | grep timing | tail -19 > "$out/genome_native_laps.txt"
# one pattern's chain alone on a rank's share (no second chain beside it)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out/kt_b" -o rs -- python tools/time_pattern_alone.py 8 1 1 > "$out/kt_b.log" 2>&1
python tools/kernel_timeline.py "$out/kt_b" narrow_enumerate > "$out/borders_alone_timeline.txt" 2>&1
python tools/time_c4_phases.py 6 > "$out/c4_phases.txt" 2>&1
python tools/time_c5_phases.py > "$out/c5_phases.txt" 2>&1
[ -f chromosight_amd/csrc/build/libchromosight_hip_prof.so ] && python tools/prof_mfma_sections.py c2 c3k c4p > "$out/tile_kernel_sections.txt" 2>&1
bash tools/c4_mode_sweep.sh "$out/c4_modes.txt" > /dev/null 2>&1

# helper kernels of the CSR path: bytes moved by the distance law / tiler (c3)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_c3_$c" -o c3 -- \
      python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > "$out/pmc_c3_$c.log" 2>&1
done

# 4. FETCH_SIZE calibration on a known byte count with the kernel's own load pattern, and the
#    micro-benchmarks behind two design decisions (FMA forms; MFMA beside packed FMAs)
mkdir -p tools/ubench/build
for u in fetch_calib coexec fma_rate write_rate; do
  [ -x tools/ubench/build/$u ] || hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o tools/ubench/build/$u 2>/dev/null
done
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/calib" -o calib -- \
    tools/ubench/build/fetch_calib > "$out/calib.log" 2>&1
python tools/time_templates.py c3 > "$out/template_kernels.txt" 2>/dev/null
python tools/time_templates.py c4p >> "$out/template_kernels.txt" 2>/dev/null
# templates with a side of 18 .. 33 (cs_corr_wide.hip): timings, per-phase cycle shares, kernel trace and counters of the
# 21 x 21 template on the dense 4096^2 map and on C4' (own passes)
python tools/time_wide.py c3 dense c4p >> "$out/template_kernels.txt" 2>/dev/null
[ -f chromosight_amd/csrc/build/libchromosight_hip_prof.so ] && python tools/prof_wide_sections.py dense c4p 21 33 > "$out/wide_kernel_sections.txt" 2>&1
for w in dense c4p; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_wide_$w" -o wide_$w -- \
      python tools/run_wide_case.py $w 21 20 > "$out/trace_wide_$w.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_wide_${w}_$c" -o wide_$w -- \
        python tools/run_wide_case.py $w 21 5 > "$out/pmc_wide_${w}_$c.log" 2>&1
  done
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
      --kernel-trace --output-format csv -d "$out/pmc_wide_${w}_mfma" -o wide_$w -- \
      python tools/run_wide_case.py $w 21 5 > "$out/pmc_wide_${w}_mfma.log" 2>&1
done
hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o tools/ubench/build/mfma_rate 2>/dev/null && tools/ubench/build/mfma_rate > "$out/ubench_mfma_rate.txt" 2>&1
tools/ubench/build/coexec > "$out/ubench_coexec.txt" 2>&1
tools/ubench/build/fma_rate > "$out/ubench_fma_rate.txt" 2>&1
tools/ubench/build/write_rate > "$out/ubench_write_rate.txt" 2>&1
ls "$out" | head -80
