#!/bin/bash
set -o pipefail
D=gpurun_out/r3s
mkdir -p $D
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x > $D/pytest_kernels.log 2>&1; echo "rc=$? kernels" >> $D/progress.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eer --no-fp32-leg > $D/bench.log 2>&1; echo "rc=$? bench" >> $D/progress.log
cat $D/progress.log; tail -3 $D/pytest_kernels.log
