#!/bin/bash
set -o pipefail
D=gpurun_out/r3e
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
V=$PWD/pytorch-kaldi-resnet_amd/variants
run bench_default 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eer --no-f16-window
SPK_LIB=$V/libspkhip_abl_B_HALF.so run bench_b_half 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eer --no-f16-window
SPK_LIB=$V/libspkhip_abl_NO_BLOAD.so run bench_no_bload 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eer --no-f16-window
(rocprofv3 -L > $D/counters.txt 2>&1 || true)
cat $D/progress.log
