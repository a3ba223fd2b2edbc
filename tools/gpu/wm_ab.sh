#!/bin/bash
# A/B inside the training step on ONE box, alternating runs, of the 16x16x32 weight gradients with dY by LDS DMA (csrc/conv_wgrad_wm16.hip):
# SPK_C32M16 = the 32-channel-group layout (first layer) against conv_wgrad_split_kernel; SPK_WM16 = the 2 x 2 layout against conv_wgrad_wm_kernel.
# WM_AB="name:VAR=val,VAR=val name2:..." overrides the plan.
D=${1:-gpurun_out/wm_ab}
mkdir -p $D
FAST="--steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-fp32-leg --no-f16-window --no-extra"
one() {
    local name=$1; shift
    echo "=== $name $(date +%T)"
    env "$@" SPK_LABEL_SHAPES=1 timeout -k 10 300 python3 bench.py $FAST > $D/$name.json 2> $D/$name.err
    local rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
    python3 - $D/$name.json $name <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("  no JSON line:", e); sys.exit(0)
print("  %s: %.2f ms/step  %.0f utt/s  loss %s -> %s" % (sys.argv[2], j["ms_per_step"], j["value"], j["first_loss"], j["final_loss"]))
for k, v in sorted(((j.get("roofline") or {}).get("all_kernels") or {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if "wgrad" in k and v["ms_per_step"] >= 0.05:
        print("    %-62s %7.3f ms %3d x %.3f" % (k[:62], v["ms_per_step"], v["launches_per_step"], v["ms_per_step"] / max(1, v["launches_per_step"])))
PY
}
for spec in ${WM_AB:-c32_a:SPK_C32M16=1 split_a:SPK_C32M16=0 c32_b:SPK_C32M16=1 split_b:SPK_C32M16=0}; do
    one ${spec%%:*} $(echo ${spec#*:} | tr ',' ' ')
done
