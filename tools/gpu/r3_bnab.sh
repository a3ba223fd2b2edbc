#!/bin/bash
set -o pipefail
D=gpurun_out/r3j
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
A="--steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-f16-window --no-fp32-leg"
run pytest_bn 400 python3 -m pytest tests/test_kernels_gpu.py tests/test_pairs_gpu.py -q -m gpu -x -k "batchnorm or bn_ or pair or bound"
run bench_u2 300 python3 bench.py $A
SPK_LIB=$V/libspkhip_bn_u1.so run bench_u1 300 python3 bench.py $A
SPK_LIB=$V/libspkhip_bn_u4.so run bench_u4 300 python3 bench.py $A
run bench_u2b 300 python3 bench.py $A
cat $D/progress.log
