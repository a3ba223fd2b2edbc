#!/bin/bash
set -o pipefail
D=gpurun_out/r3w
mkdir -p $D
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $D/bench_final.log 2> $D/bench_final.err; echo "rc=$? final" >> $D/progress.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-roofline --no-f16-window --ingest > $D/bench_ingest.log 2> $D/bench_ingest.err; echo "rc=$? ingest" >> $D/progress.log
cat $D/progress.log
