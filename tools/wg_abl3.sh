#!/bin/bash
# ablations of conv_wgrad_ws_kernel: NO_K consumers idle, NO_P no publish, NO_L no global loads (tools/variant.sh ... -DWGWS_ABL_*)
for v in "" NO_K NO_P NO_L NO_PL; do
  if [ -z "$v" ]; then L=""; else L="SPK_LIB=pytorch-kaldi-resnet_amd/variants/libspkhip_wgws_$v.so"; fi
  echo "== variant ${v:-full}"
  env $L timeout -k 10 200 python tools/conv_bench.py --reps 5 2>&1 | grep -E "wgrad" | sed -E 's/.*(wgrad [0-9.]+ ms +[0-9.]+ TF).*/\1/' | paste -sd' '
done
