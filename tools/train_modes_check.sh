#!/bin/bash
# Same short training run (synthetic corpus, 10 speakers, seeded) with fp32 operands and with the default bf16-split operands:
# prints the per-epoch validation lines of both for a side-by-side look at the training dynamics.  GPU box only.
set -e
D=${1:-/tmp/tmc}
LOSS=${LOSS:-AAM}
EPOCHS=${EPOCHS:-4}
LR=${LR:-0.05}
MODES=${MODES:-"f32 bf16x6 f16x3"}
python tools/make_synth_data.py --out $D --speakers 10 --utts-per-speaker 100 --min-frames 200 --max-frames 260 --trials 2000 > /dev/null
for mode in $MODES; do
  SPK_MFMA=$mode python scripts/train_resnet.py --train-list $D/train.scp --cv-list $D/cv.scp --utt2spkid $D/utt2spkid \
    --input-dim 80 --spk-num 10 --pooling mean+std --loss-type $LOSS --min-chunk-size 200 --max-chunk-size 200 \
    --log-dir $D/exp_$mode --arch resnet34 --epochs $EPOCHS -b 32 --lr $LR --lr-final 0.001 --wd 5e-4 -p 1000 --seed 7 --gpu 0 \
    --native-reader -j 4 > $D/log_$mode.txt 2>&1 || { echo "== $mode FAILED"; tail -20 $D/log_$mode.txt; continue; }
  echo "== $mode"; grep -E "^ \* Acc@1" $D/log_$mode.txt | tr "\n" " "; echo; grep -E "train throughput" $D/log_$mode.txt | tail -1
done
