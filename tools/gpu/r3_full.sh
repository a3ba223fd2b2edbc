#!/bin/bash
set -o pipefail
D=gpurun_out/r3full
mkdir -p $D
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $D/pytest_gpu.log 2>&1; rc=$?; echo "rc=$rc pytest" >> $D/progress.log
tail -5 $D/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
