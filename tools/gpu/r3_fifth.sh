#!/bin/bash
set -o pipefail
D=gpurun_out/r3f
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
run bench_c4 400 python3 bench.py --config c4 --steps 16 --warmup 8 --no-cpu-baseline --no-eer
SPK_FUSE_APPLY_MAXC=1000000 run bench_c4_fused 400 python3 bench.py --config c4 --steps 16 --warmup 8 --no-cpu-baseline --no-eer
run bench_ingest 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-roofline --ingest
SPK_FORCE_DEVICE=0 SPK_DIST_BACKEND=gloo run bench_gpus2 400 python3 bench.py --gpus 2 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline --no-eer --no-roofline
run bench_c5 300 python3 bench.py --config c5 --steps 20 --warmup 5
run pytest_pipeline 900 python3 -m pytest tests/test_pipeline_gpu.py -q -m gpu -x -s
cat $D/progress.log
