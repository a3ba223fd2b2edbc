#!/bin/bash
set -o pipefail
D=gpurun_out/r3l
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
A="--steps 10 --warmup 3 --no-cpu-baseline --no-eer --no-f16-window --no-fp32-leg"
run base 300 python3 bench.py $A
SPK_FUSE_APPLY_MAXC=0 run unfused_all 300 python3 bench.py $A
run base2 300 python3 bench.py $A
SPK_FUSE_APPLY_MAXC=0 run unfused_all2 300 python3 bench.py $A
cat $D/progress.log
