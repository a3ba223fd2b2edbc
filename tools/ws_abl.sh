#!/bin/bash
# Ablation builds of conv_ws_kernel (wrong results by construction) timed on the ResNet-34 layer shapes: what each phase costs.
for v in "" NO_BLOAD NO_STAGE NO_EPI NO_SYNC ALL; do
  if [ -z "$v" ]; then L=""; else L="SPK_LIB=pytorch-kaldi-resnet_amd/variants/libspkhip_ws_$v.so"; fi
  echo "== variant ${v:-full}"
  env $L timeout -k 10 200 python tools/ws_check.py --reps 4 2>&1 | grep -E "^    (fwd|dgrad)|^L[0-9]" | sed 's/bit-equal.*//'
done
