#!/bin/bash
set -o pipefail
D=gpurun_out/r3z
mkdir -p $D
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "16x16x32 or pipelined_conv_equals" > $D/t_m16.log 2>&1; rc=$?; echo "rc=$rc m16 tests" >> $D/progress.log
tail -25 $D/t_m16.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
  for v in 0 1; do
    SPK_PIPE_M16=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-f16-window --no-fp32-leg > $D/b_${v}_$i.log 2> $D/b_${v}_$i.err; rc=$?
    echo "rc=$rc m16=$v $i $(python3 -c "import json,sys; d=json.loads([l for l in open('$D/b_${v}_$i.log') if l.startswith('{')][-1]); k=d['roofline']['all_kernels']; print(d['ms_per_step'], d['final_loss'], {n:v['ms_per_step'] for n,v in k.items() if 'pipe' in n})")" >> $D/progress.log
    [ $rc -eq 0 ] || exit 1
  done
done
cat $D/progress.log
