#!/bin/bash
# Ablation builds of conv_mfma_kernel<.., 3> (f16x3; wrong results by construction; tools/variant.sh cv_* conv_split.hip -DABL_*)
# timed on the ResNet-34 layer shapes: forward and plain data gradient per launch shape.
for v in "" NO_STAGE NO_BLOAD NO_ALOAD NO_EPI K KM; do
  if [ -z "$v" ]; then L=""; else L="SPK_LIB=pytorch-kaldi-resnet_amd/variants/libspkhip_cv_$v.so"; fi
  echo "== variant ${v:-full}"
  env $L SPK_CONV_PIPE=0 timeout -k 10 200 python tools/conv_bench.py --reps 5 2>&1 | grep -E "3x3 s1" | sed -E 's/^(L[0-9]).*(fwd [0-9.]+ ms +[0-9.]+ TF) +(dgrad [0-9.]+ ms +[0-9.]+ TF).*/\1 \2 \3/' | paste -sd'|'
done
