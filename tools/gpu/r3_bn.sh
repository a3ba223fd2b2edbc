#!/bin/bash
set -o pipefail
D=gpurun_out/r3bn
mkdir -p $D
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "bn or batchnorm" > $D/t_bn.log 2>&1; rc=$?; echo "rc=$rc bn tests" >> $D/progress.log
[ $rc -eq 0 ] || { tail -30 $D/t_bn.log; exit 1; }
for i in 1 2; do
  for v in new s2048 s1024 s8192; do
    if [ $v = new ]; then unset SPK_LIB; else export SPK_LIB=$PWD/pytorch-kaldi-resnet_amd/variants/libspkhip_bn_$v.so; fi
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-f16-window --no-fp32-leg > $D/b_${v}_$i.log 2> $D/b_${v}_$i.err; rc=$?
    echo "rc=$rc $v $i $(python3 -c "import json,sys; d=json.loads([l for l in open('$D/b_${v}_$i.log') if l.startswith('{')][-1]); k=d['roofline']['all_kernels']; print(d['ms_per_step'], d['final_loss'], {n:v['ms_per_step'] for n,v in k.items() if v['ms_per_step']>2.0})")" >> $D/progress.log
    [ $rc -eq 0 ] || exit 1
  done
done
cat $D/progress.log
