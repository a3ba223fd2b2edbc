#!/bin/bash
set -o pipefail
D=gpurun_out/r3d
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
run bench_default 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer
SPK_LIB=$PWD/pytorch-kaldi-resnet_amd/variants/libspkhip_epi0.so run bench_epi0 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer
run bench_default2 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer
run pytest_kernels 700 python3 -m pytest tests/test_kernels_gpu.py tests/test_pairs_gpu.py -q -m gpu -x
cat $D/progress.log
