#!/bin/bash
# a short real training run through the entry point with the final kernels (f16x3 and the exact bf16x6 mode side by side), then
# the window counters / same-forward backward distances on the trained checkpoint
set -o pipefail
D=gpurun_out/r3t
mkdir -p $D
(EPOCHS=8 LR=0.01 MODES="f16x3 bf16x6" timeout -k 10 900 bash tools/train_modes_check.sh /tmp/tmc > $D/train_modes.log 2>&1; echo "rc=$? train" >> $D/progress.log)
timeout -k 10 300 python3 tools/window_on_checkpoint.py /tmp/tmc/exp_f16x3/checkpoint_epoch7.pth.tar /tmp/tmc/train.scp /tmp/tmc/utt2spkid 64 200 4 > $D/windows_trained.log 2>&1; echo "rc=$? windows" >> $D/progress.log
cat $D/progress.log; cat $D/train_modes.log; tail -25 $D/windows_trained.log
rm -rf /tmp/tmc
