#!/bin/bash
set -o pipefail
D=gpurun_out/r3b
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
run pytest_pairs 300 python3 -m pytest tests/test_pairs_gpu.py -q -m gpu -s
run pytest_fullsize 400 python3 -m pytest tests/test_fullsize_gpu.py -q -m gpu -s
run bench_default 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer
SPK_PAIR_DRAW=0 SPK_FUSE_APPLY_MAXC=1000000 run bench_r02_policy 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer
run pytest_kernels 600 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x
run pytest_model 500 python3 -m pytest tests/test_model_gpu.py -q -m gpu -x
cat $D/progress.log
