#!/bin/bash
# A/B of variant libraries inside the training step on ONE box (tools/variant.sh builds them):
#   usage (via run.sh): sh:tools/gpu/ab.sh   with AB_VARIANTS="base u3 nocvt" [AB_ARGS="--config c4 ..."] [AB_GREP=regex of kernel labels]
# per variant: the step time of the labelled bench (SPK_LABEL_SHAPES=1, eager instrumented pass + graph replay time)
D=${1:-gpurun_out/ab}
mkdir -p $D
for v in ${AB_VARIANTS:-base}; do
    lib=pytorch-kaldi-resnet_amd/variants/libspkhip_$v.so
    [ "$v" = base ] && lib=pytorch-kaldi-resnet_amd/libspkhip.so
    echo "=== $v $(date +%T)"
    SPK_LIB=$lib SPK_LABEL_SHAPES=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eer --no-fp32-leg --no-f16-window --no-extra ${AB_ARGS} > $D/ab_$v.json 2> $D/ab_$v.err
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $v: stopping"; exit 1; fi
    python3 - $D/ab_$v.json "$v" "${AB_GREP:-.}" <<'PY'
import json, re, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("  no JSON line:", e); sys.exit(0)
print("  %s: %.2f ms/step  %.0f utt/s  loss %s -> %s" % (sys.argv[2], j["ms_per_step"], j["value"], j["first_loss"], j["final_loss"]))
r = j.get("roofline") or {}
for k, v in sorted((r.get("all_kernels") or {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if re.search(sys.argv[3], k) and v["ms_per_step"] >= 0.05:
        print("    %-62s %7.3f ms %3d x %.3f" % (k[:62], v["ms_per_step"], v["launches_per_step"], v["ms_per_step"] / max(1, v["launches_per_step"])))
PY
done
