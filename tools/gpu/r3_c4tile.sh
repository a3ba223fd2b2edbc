#!/bin/bash
set -o pipefail
D=gpurun_out/r3u
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
A="--arch resnet101 --speakers 5994 --frames 300 --steps 6 --warmup 3 --no-cpu-baseline --no-eer --no-f16-window"
run base 300 python3 bench.py $A
for t in 5,25,1,4 4,32,1,4 10,25,2,2 10,25,2,4 15,25,3,2 5,25,1,1 20,25,4,2 3,25,1,2 2,32,1,2; do
  SPK_PLAIN_TILE="20,75,1,128:$t" SPK_LABEL_SHAPES=1 run t_${t//,/_} 300 python3 bench.py $A
done
cat $D/progress.log
