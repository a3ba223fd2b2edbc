#!/bin/bash
# round-3 first GPU call: probe, new parity tests, three A/B bench runs of the BatchNorm-backward policy
set -o pipefail
mkdir -p gpurun_out/r3a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
run() {   # name, timeout, command...: stop the whole call after a timeout / kill
    local name=$1 to=$2; shift 2
    echo "=== $name" | tee -a gpurun_out/r3a/progress.log
    timeout -k 10 "$to" "$@" > gpurun_out/r3a/$name.log 2>&1
    local rc=$?
    echo "rc=$rc $name" | tee -a gpurun_out/r3a/progress.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/r3a/progress.log; exit 1; fi
    return 0
}
(cd tools/probe && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC split_probe.hip -o split_probe.so) > gpurun_out/r3a/probe_build.log 2>&1
run split_probe 300 python3 tools/probe/run_split_probe.py
run pytest_pairs 900 python3 -m pytest tests/test_pairs_gpu.py -x -q -m gpu -s
run pytest_fullsize 900 python3 -m pytest tests/test_fullsize_gpu.py -q -m gpu -s
run pytest_parallel 900 python3 -m pytest tests/test_parallel_gpu.py -q -m gpu -s
run bench_default 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer
SPK_FUSE_APPLY_MAXC=1000000 run bench_fused_pairs 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer
SPK_PAIR_DRAW=0 SPK_FUSE_APPLY_MAXC=1000000 run bench_r02_policy 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer
tail -3 gpurun_out/r3a/*.log | tail -80
