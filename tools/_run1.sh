o=gpurun_out/s22; mkdir -p $o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d $o/p1 -o p -- python tools/time_pattern_alone.py 8 1 1 > $o/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $o/p2 -o p -- python tools/time_pattern_alone.py 8 1 1 > $o/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/s22/p1", "gpurun_out/s22/p2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "rescore_run_batch" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
