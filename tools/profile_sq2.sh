#!/bin/bash
# Second SQ counter pass of the default bench command (GPU box): what the waves of the matrix-bound kernels wait on - LDS issue
# stalls, LDS instructions / array cycles / bank conflicts, vector-memory instructions and their level, matrix and vector
# instruction counts, co-execution cycles.  Two --pmc passes (8 SQ slots each), never combined with a trace domain.
# usage: tools/profile_sq2.sh <outdir> [bench args]
set -e
OUT=$(readlink -f "$1"); shift
R=$(readlink -f "$(dirname "$0")/..")
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-eer --no-roofline --no-f16-window --no-extra $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU \
  -d "$OUT/pmc_a" --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/bench_line_a.json" 2> "$OUT/a.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY \
  -d "$OUT/pmc_b" --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/bench_line_b.json" 2> "$OUT/b.err"
cd "$R"
python3 tools/pmc_sq_summary.py "$OUT/pmc_a" 16 > "$OUT/sq_counters_lds_vmem.md"
python3 tools/pmc_sq_summary.py "$OUT/pmc_b" 16 > "$OUT/sq_counters_issue.md"
python3 tools/pmc_sq_summary.py "$OUT/pmc_a" 40 --by-grid > "$OUT/sq_counters_lds_vmem_by_grid.md"
python3 tools/pmc_sq_summary.py "$OUT/pmc_b" 40 --by-grid > "$OUT/sq_counters_issue_by_grid.md"
rm -rf "$OUT/pmc_a" "$OUT/pmc_b"
cut -c1-260 "$OUT/sq_counters_lds_vmem.md" "$OUT/sq_counters_issue.md"
