#!/bin/bash
# Launch time of the streaming 32-channel 3x3 kernel under each diagnostic build in LIBS (names of tools/variant.sh libraries;
# "base" = the regular library), on one box.  usage (inside gpurun): LIBS="base c32stamps ..." bash tools/gpu/c32_abl.sh <outdir>
D=${1:-gpurun_out/c32abl}
mkdir -p $D
for name in ${LIBS:-base}; do
    echo "=== $name $(date +%T)" | tee -a $D/c32_abl.log
    if [ "$name" = base ]; then unset SPK_LIB; else export SPK_LIB=$(pwd)/pytorch-kaldi-resnet_amd/variants/libspkhip_$name.so; fi
    timeout -k 10 200 python3 ${C32_TOOL:-tools/c32_stamps.py} >> $D/c32_abl.log 2>> $D/c32_abl.err
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
done
cat $D/c32_abl.log
