#!/bin/bash
# usage: pmc_one.sh <tag> [SPK_LIB=...]   -> per-kernel SQ counters + durations of tools/ws_check.py --layers 4
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
env "$@" timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_$TAG --output-format csv -- python3 $R/tools/ws_check.py --layers 4 --reps 3 > $R/gpurun_out/pmc_$TAG.log 2>&1
env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$TAG --output-format csv -- python3 $R/tools/ws_check.py --layers 4 --reps 3 > $R/gpurun_out/kt_$TAG.log 2>&1
cd $R
python3 tools/pmc_sq_summary.py gpurun_out/pmc_$TAG 4 | cut -c1-260
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/kt_$TAG/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_" in r["Name"]:
            print(r["Name"].split("(")[0][:60], r["Calls"], "avg_us", float(r["AverageNs"])/1e3)
PY
