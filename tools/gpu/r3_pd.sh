#!/bin/bash
set -o pipefail
D=gpurun_out/r3pd
mkdir -p $D
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "16x16x32" > $D/t.log 2>&1; rc=$?; echo "rc=$rc m16 tests" >> $D/progress.log
[ $rc -eq 0 ] || { tail -30 $D/t.log; exit 1; }
for i in 1 2; do
  for v in new d4c3 d4c4 d5c3; do
    if [ $v = new ]; then unset SPK_LIB; else export SPK_LIB=$PWD/pytorch-kaldi-resnet_amd/variants/libspkhip_p_$v.so; fi
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-f16-window --no-fp32-leg > $D/b_${v}_$i.log 2> $D/b_${v}_$i.err; rc=$?
    echo "rc=$rc $v $i $(python3 -c "import json,sys; d=json.loads([l for l in open('$D/b_${v}_$i.log') if l.startswith('{')][-1]); k=d['roofline']['all_kernels']; print(d['ms_per_step'], d['final_loss'], {n:v['ms_per_step'] for n,v in k.items() if 'pipe_kernel<3' in n})")" >> $D/progress.log
    [ $rc -eq 0 ] || exit 1
  done
done
cat $D/progress.log
