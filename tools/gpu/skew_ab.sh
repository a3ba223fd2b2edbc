#!/bin/bash
# A/B of the row-skew variant of the weight gradient's X image (tools/variant.sh skew conv_wgrad_wm16.hip -DWM16_ROW_SKEW=192) on one box:
# its own tests against the variant library, then alternating bench runs on the ResNet-34 step and on the ResNet-101 step.
D=${1:-gpurun_out/skew_ab}
mkdir -p $D
V=pytorch-kaldi-resnet_amd/variants/libspkhip_skew.so
echo "=== tests against the variant $(date +%T)"
SPK_LIB=$V timeout -k 10 150 python3 -m pytest tests -m gpu -x -q -k "weight_gradient_16x16x32" > $D/skew_tests.log 2>&1
echo "rc=$? $(tail -1 $D/skew_tests.log)"
AB_VARIANTS="base skew base skew" AB_GREP="wgrad_(wm16|c32m16)" bash tools/gpu/ab.sh $D
AB_VARIANTS="base skew" AB_ARGS="--config c4 --lengths 3" AB_GREP="wgrad_wm16" bash tools/gpu/ab.sh $D/c4
