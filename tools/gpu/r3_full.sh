#!/bin/bash
set -o pipefail
D=gpurun_out/r3g
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
run bench_c4 400 python3 bench.py --config c4 --steps 16 --warmup 8 --no-cpu-baseline --no-eer
run pytest_all 1000 python3 -m pytest tests -q -m gpu
run smoke 200 python3 __graft_entry__.py smoke
cat $D/progress.log
