#!/bin/bash
# BASELINE configs[0] plumbing run (the reference's run_aam_cpu.sh shape: ResNet-34 + AAM, 1k synthetic
# 200-frame x 80 fbank utts, 10 speakers, bs 32) - on this stack the "cpu" recipe still trains on the MI355X
# (there is no CPU compute path); the CPU leg is the oracle timed by bench.py.
# usage: ./run_aam_cpu.sh <work-dir>          (SPK_SEED=<int> seeds the training run; the reference's recipe passes no seed, and
# neither does this one by default: 60 SGD steps from a random initialisation end anywhere between 12 % and 37 % EER, measured)
set -e
dir=${1:-exp/aam_c1}
mkdir -p $dir/data
python tools/make_synth_data.py --out $dir/data --speakers 10 --utts-per-speaker 100 --min-frames 200 --max-frames 260 --feat-dim 80
python scripts/train_resnet.py --gpu 0 --workers 4 --batch-size 32 --print-freq 10 \
  --arch resnet34 --input-dim 80 --loss-type AAM --pooling 'mean+std' --margin 0.2 --scale 30 \
  --dataset v1 --epochs 2 --lr 0.01 --lr-final 0.0001 --wd 5e-4 --min-chunk-size 200 --max-chunk-size 200 \
  --train-list $dir/data/train.scp --cv-list $dir/data/cv.scp --spk-num 10 --utt2spkid $dir/data/utt2spkid \
  --log-dir $dir ${SPK_SEED:+--seed $SPK_SEED} | tee $dir/train.log
model=$dir/checkpoint_epoch1.pth.tar
python scripts/decode.py --gpu 0 --workers 2 --batch-size 1 --chunk-size -1 --spk_num 10 --arch resnet34 \
  --input-dim 80 --pooling 'mean+std' --model-path $model --decode-scp $dir/data/all.scp --out-path $dir/embeddings
python scripts/compute_mean.py $dir/embeddings/alone $dir/mean.vec
python scripts/cosine_score.py --mean $dir/mean.vec --enroll $dir/embeddings/alone --test $dir/embeddings/alone \
  --trials $dir/data/trials --score-file $dir/scores
echo "EER: $(python scripts/compute_eer.py $dir/scores $dir/data/trials 2>/dev/null)" | tee $dir/eer_cosine
