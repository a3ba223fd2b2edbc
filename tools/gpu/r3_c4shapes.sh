#!/bin/bash
D=gpurun_out/r3t
mkdir -p $D
SPK_LABEL_SHAPES=1 timeout -k 10 400 python3 bench.py --config c4 --steps 8 --warmup 8 --no-cpu-baseline --no-eer --no-f16-window > $D/c4_shapes.log 2> $D/c4_shapes.err; echo rc=$?
