#!/bin/bash
# Board power and clocks while the training step replays (is the step running at the chip's power limit?).
# Runs the bench line in the background for ~25 s of replays and samples rocm-smi twice a second; no GPU context of its own.
#   usage (inside gpurun): bash tools/gpu/power_trace.sh <outdir> [bench args]
D=${1:-gpurun_out/power}; shift
mkdir -p $D
rocm-smi --showpower --showclocks --showmaxpower --showtemp > $D/idle.txt 2>&1
timeout -k 10 280 python3 bench.py --steps ${POWER_STEPS:-500} --warmup 5 --no-cpu-baseline --no-eer --no-fp32-leg --no-f16-window --no-extra --no-roofline "$@" > $D/bench.json 2> $D/bench.err &
BP=$!
: > $D/samples.txt
for i in $(seq 1 ${POWER_SAMPLES:-90}); do
    if ! kill -0 $BP 2>/dev/null; then break; fi
    echo "--- $(date +%T.%N)" >> $D/samples.txt
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" >> $D/samples.txt
    sleep 0.4
done
wait $BP
echo "bench rc=$?"
tail -c 400 $D/bench.json
python3 - $D/samples.txt <<'PY'
import re, sys
pw, sc = [], []
for line in open(sys.argv[1]):
    m = re.search(r"Power \(W\): ([0-9.]+)", line)
    if m: pw.append(float(m.group(1)))
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", line)
    if m: sc.append(int(m.group(1)))
print("samples %d; power W: %s" % (len(pw), " ".join("%.0f" % p for p in pw)))
print("sclk MHz: %s" % " ".join(str(s) for s in sc))
PY
