#!/bin/bash
set -o pipefail
D=gpurun_out/r3q
mkdir -p $D
(EPOCHS=${EPOCHS:-8} LR=0.01 MODES="f16x3" timeout -k 10 600 bash tools/train_modes_check.sh /tmp/tmc > $D/train_modes.log 2>&1; echo "rc=$? train" >> $D/progress.log)
timeout -k 10 300 python3 tools/window_on_checkpoint.py /tmp/tmc/exp_f16x3/checkpoint_epoch$((${EPOCHS:-8}-1)).pth.tar /tmp/tmc/train.scp /tmp/tmc/utt2spkid 64 200 3 > $D/windows_trained.log 2>&1; echo "rc=$? windows" >> $D/progress.log
cat $D/progress.log; tail -3 $D/train_modes.log; cat $D/windows_trained.log | tail -14
rm -rf /tmp/tmc
