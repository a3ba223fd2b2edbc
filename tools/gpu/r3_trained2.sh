#!/bin/bash
set -o pipefail
D=gpurun_out/r3r
mkdir -p $D
timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -m gpu -s -k "stats_pool or dead_channel" > $D/pytest.log 2>&1; echo "rc=$? pytest" >> $D/progress.log
(EPOCHS=8 LR=0.01 MODES="f16x3" timeout -k 10 600 bash tools/train_modes_check.sh /tmp/tmc > $D/train_modes.log 2>&1; echo "rc=$? train" >> $D/progress.log)
timeout -k 10 300 python3 tools/window_on_checkpoint.py /tmp/tmc/exp_f16x3/checkpoint_epoch7.pth.tar /tmp/tmc/train.scp /tmp/tmc/utt2spkid 64 200 4 > $D/windows_trained.log 2>&1; echo "rc=$? windows" >> $D/progress.log
cat $D/progress.log
rm -rf /tmp/tmc
