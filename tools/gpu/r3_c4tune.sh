#!/bin/bash
set -o pipefail
D=gpurun_out/r3n
mkdir -p $D
run() {
    local name=$1 to=$2; shift 2
    echo "=== $name $(date +%T)" | tee -a $D/progress.log
    timeout -k 10 "$to" "$@" > $D/$name.log 2> $D/$name.err
    local rc=$?
    echo "rc=$rc $name" | tee -a $D/progress.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $D/progress.log; exit 1; fi
    return 0
}
A="--config c4 --steps 16 --warmup 8 --no-cpu-baseline --no-eer --no-f16-window"
run c4_base 400 python3 bench.py $A
run c4_autotune 900 python3 bench.py $A --autotune
run pytest_model 600 python3 -m pytest tests/test_model_gpu.py -q -m gpu -x -k "forward_parity or backward_parity"
cat $D/progress.log
