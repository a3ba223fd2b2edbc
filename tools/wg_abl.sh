#!/bin/bash
# Ablation builds of conv_wgrad_split_kernel (wrong results by construction; tools/variant.sh ... -DWG_ABL_*) timed on the
# ResNet-34 layer shapes (tools/conv_bench.py): what each phase of the kernel costs.
for v in "" NO_PREFETCH NO_PUBLISH NO_KLOOP NO_MMA; do
  if [ -z "$v" ]; then L=""; else L="SPK_LIB=pytorch-kaldi-resnet_amd/variants/libspkhip_wg_$v.so"; fi
  echo "== variant ${v:-full}"
  env $L timeout -k 10 200 python tools/conv_bench.py --reps 5 2>&1 | grep -E "wgrad" | sed -E 's/.*(wgrad [0-9.]+ ms +[0-9.]+ TF).*/\1/' | paste -sd' '
done
