#!/bin/bash
set -o pipefail
D=gpurun_out/r3l1
mkdir -p $D
for i in 1 2; do
  for v in 32 16; do
    SPK_FUSE_APPLY_MAXC=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-f16-window --no-fp32-leg > $D/b_${v}_$i.log 2> $D/b_${v}_$i.err; rc=$?
    echo "rc=$rc maxc=$v $i $(python3 -c "import json,sys; d=json.loads([l for l in open('$D/b_${v}_$i.log') if l.startswith('{')][-1]); k=d['roofline']['all_kernels']; print(d['ms_per_step'], d['final_loss'], {n:(v['ms_per_step'],v['launches_per_step']) for n,v in k.items() if v['ms_per_step']>2.5})")" >> $D/progress.log
    [ $rc -eq 0 ] || exit 1
  done
done
cat $D/progress.log
