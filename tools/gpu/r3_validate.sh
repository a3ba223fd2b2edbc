#!/bin/bash
set -o pipefail
D=gpurun_out/r3v
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
run pytest_all 1000 python3 -m pytest tests -q -m gpu
run smoke 200 python3 __graft_entry__.py smoke
cat $D/progress.log; tail -4 $D/pytest_all.log; tail -2 $D/smoke.log
