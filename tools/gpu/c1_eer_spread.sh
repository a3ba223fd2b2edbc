#!/bin/bash
# Spread of the EER that ./run_aam_cpu.sh ends with (BASELINE configs[0]: 60 SGD steps from a random initialisation, NO seed - as the
# reference's recipe passes none): N runs per library.  usage (via run.sh): sh:tools/gpu/c1_eer_spread.sh  [C1_LIBS="base oldpitch"] [C1_RUNS=4]
D=${1:-gpurun_out/c1_eer}
mkdir -p $D
for v in ${C1_LIBS:-base}; do
    lib=pytorch-kaldi-resnet_amd/variants/libspkhip_$v.so
    [ "$v" = base ] && lib=pytorch-kaldi-resnet_amd/libspkhip.so
    for i in $(seq 1 ${C1_RUNS:-4}); do
        w=$(mktemp -d /tmp/c1_XXXXXX)
        SPK_LIB=$lib PYTHONPATH=$(pwd) timeout -k 10 200 bash run_aam_cpu.sh $w > $D/c1_${v}_$i.log 2> $D/c1_${v}_$i.err
        rc=$?
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $v run $i: stopping"; exit 1; fi
        echo "$v run $i rc=$rc: $(cat $w/eer_cosine 2>/dev/null)  $(grep -o ' \* Acc@1 [0-9.]*' $w/train.log | tail -1)"
        rm -rf $w
    done
done
