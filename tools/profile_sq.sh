#!/bin/bash
# SQ counter pass of the default bench command (GPU box): wave cycles, wait buckets, matrix-pipe busy cycles per kernel.
set -e
OUT=$(readlink -f "$1"); shift
R=$(readlink -f "$(dirname "$0")/..")
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  -d "$OUT/pmc_sq" --output-format csv -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-eer --no-roofline --no-f16-window --no-extra "$@" > "$OUT/bench_line_sq.json" 2> "$OUT/sq.err"
cd "$R"
python3 tools/pmc_sq_summary.py "$OUT/pmc_sq" 16 > "$OUT/sq_counters.md"
cat "$OUT/sq_counters.md" | cut -c1-250
