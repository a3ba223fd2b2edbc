#!/bin/bash
# rocprofv3 passes of the default bench command (GPU box): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes (they do not fit one pass; never combined with a trace domain).  Usage: tools/profile_bench.sh <outdir> [bench args]
set -e
OUT=$(readlink -f "$1"); shift
R=$(readlink -f "$(dirname "$0")/..")
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-eer --no-fp32-leg --no-f16-window --no-extra $*"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/bench_line_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" --output-format csv -- python3 "$R/bench.py" $ARGS --no-roofline > "$OUT/bench_line_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" --output-format csv -- python3 "$R/bench.py" $ARGS --no-roofline > "$OUT/bench_line_write.json" 2> "$OUT/write.err"
cd "$R"
python3 tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic.json"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-70s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"].split("(")[0].replace("void ", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
