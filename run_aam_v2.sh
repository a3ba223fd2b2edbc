#!/bin/bash
# Reference run_aam_v2.sh stages 7-10 on one 8-GPU MI355X node (softmax pre-train -> AAM -> decode -> cosine EER),
# same flags; data dir must hold train_orig.scp / cv_orig.scp / utt2spkid / decode_{train,test}.scp / trials.
# usage: ./run_aam_v2.sh <datadir> <expdir> <num_spk> [input_dim]
set -e
datadir=$1; dir=$2; num_spk=$3; dim=${4:-40}
epoch=30; lr=0.1; lr_final=0.0; wd=5e-4; margin=0.2; scale=30
common="--multiprocessing-distributed --world-size 1 --rank 0 --gpu-num 8 --workers 16 --batch-size 1024 --print-freq 500 \
  --dist-url tcp://127.0.0.1:27544 --arch resnet34 --input-dim $dim --pooling mean+std --dataset v1 --epochs $epoch \
  --lr $lr --lr-final $lr_final --wd $wd --min-chunk-size 200 --max-chunk-size 200 \
  --train-list $datadir/train_orig.scp --cv-list $datadir/cv_orig.scp --spk-num $num_spk --utt2spkid $datadir/utt2spkid"
mkdir -p $dir/pretrained $dir/log
python scripts/train_resnet.py $common --loss-type softmax --log-dir $dir/pretrained > $dir/log/pretrain.log
python scripts/train_resnet.py $common --loss-type AAM --margin $margin --scale $scale \
  --pretrained $dir/pretrained/model_best.pth.tar --log-dir $dir > $dir/log/train.log
for x in train test; do
  python scripts/decode.py --multiprocessing-distributed --dist-url tcp://127.0.0.1:27544 --world-size 1 --rank 0 \
    --gpu-num 8 --workers 16 --batch-size 8 --chunk-size -1 --spk_num $num_spk --arch resnet34 --input-dim $dim \
    --pooling mean+std --model-path $dir/model_best.pth.tar --decode-scp $datadir/decode_${x}.scp \
    --out-path $dir/embeddings_$x > $dir/log/decode_${x}.log
  cat $dir/embeddings_$x/* | awk '!seen[$1]++' > $dir/${x}.iv      # drop the sampler's padding duplicates
done
python scripts/compute_mean.py $dir/train.iv $dir/mean.vec
python scripts/cosine_score.py --mean $dir/mean.vec --enroll $dir/test.iv --test $dir/test.iv --trials $datadir/trials \
  --score-file $dir/scores_cosine
echo "EER: $(python scripts/compute_eer.py $dir/scores_cosine $datadir/trials 2>/dev/null)" | tee $dir/eer_cosine
