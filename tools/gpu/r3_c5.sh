#!/bin/bash
set -o pipefail
D=gpurun_out/r3i
mkdir -p $D
run() {
    local name=$1 to=$2; shift 2
    echo "=== $name $(date +%T)" | tee -a $D/progress.log
    timeout -k 10 "$to" "$@" > $D/$name.log 2>&1
    local rc=$?
    echo "rc=$rc $name" | tee -a $D/progress.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $D/progress.log; exit 1; fi
    return 0
}
run bench_default 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer
run pytest_kernels 400 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "stats_pool or batchnorm or bn_finalize"
run c5_100k 700 python3 tools/c5_extract.py --dir /tmp/c5 --speakers 1000 --utts-per-speaker 100 --frames 300 --batch 512 --trials 100000
rm -rf /tmp/c5
cat $D/progress.log
