#!/bin/bash
set -o pipefail
D=gpurun_out/r3h
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
A="--config c4 --steps 16 --warmup 8 --no-cpu-baseline --no-eer --no-roofline --no-f16-window"
run c4_hybrid 400 python3 bench.py $A
SPK_FUSE_APPLY_1X1=0 run c4_unfused 400 python3 bench.py $A
SPK_FUSE_APPLY_MAXC=1000000 run c4_fused 400 python3 bench.py $A
run c4_hybrid2 400 python3 bench.py $A
SPK_FUSE_APPLY_1X1=0 run c4_unfused2 400 python3 bench.py $A
cat $D/progress.log
