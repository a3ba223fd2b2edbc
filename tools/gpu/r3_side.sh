#!/bin/bash
set -o pipefail
D=gpurun_out/r3side
mkdir -p $D
for i in 1 2; do
  for v in 0 1; do
    SPK_GRAPH_SIDE=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-f16-window --no-fp32-leg --no-roofline > $D/b_${v}_$i.log 2> $D/b_${v}_$i.err; rc=$?
    echo "rc=$rc side=$v $i $(python3 -c "import json,sys; d=json.loads([l for l in open('$D/b_${v}_$i.log') if l.startswith('{')][-1]); print(d['ms_per_step'], d['final_loss'], d['config'].get('launch'))")" >> $D/progress.log
    [ $rc -eq 0 ] || { tail -5 $D/b_${v}_$i.err; exit 1; }
  done
done
cat $D/progress.log
