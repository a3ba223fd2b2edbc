#!/bin/bash
set -o pipefail
D=gpurun_out/r3k
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
for t in 16,15,2,1 8,32,2,1 16,24,3,1 12,32,3,1 20,19,3,1 16,32,4,1 10,25,2,1 20,12,2,1; do
  SPK_FUSED_TILE="80,300,32:$t" run tile_${t//,/_} 300 python3 bench.py $A
done
run base2 300 python3 bench.py $A
cat $D/progress.log
