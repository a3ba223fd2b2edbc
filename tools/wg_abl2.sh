#!/bin/bash
for v in K KM KL B; do
  echo "== variant $v"
  env SPK_LIB=pytorch-kaldi-resnet_amd/variants/libspkhip_wg_$v.so timeout -k 10 200 python tools/conv_bench.py --reps 5 2>&1 | grep -E "wgrad" | sed -E 's/.*(wgrad [0-9.]+ ms +[0-9.]+ TF).*/\1/' | paste -sd' '
done
