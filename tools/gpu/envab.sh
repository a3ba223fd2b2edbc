#!/bin/bash
# A/B of environment settings inside the training step on ONE box: ENVAB="name1:VAR=val,VAR2=val name2:..." (name "base" = no variables)
D=${1:-gpurun_out/envab}
mkdir -p $D
for spec in ${ENVAB:-base:}; do
    name=${spec%%:*}; vars=${spec#*:}
    echo "=== $name ($vars) $(date +%T)"
    env $(echo $vars | tr ',' ' ') timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-fp32-leg --no-f16-window --no-extra --no-roofline ${AB_ARGS} > $D/env_$name.json 2> $D/env_$name.err
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
    python3 -c "
import json,sys
try:
    j=json.loads(open('$D/env_$name.json').read().strip().splitlines()[-1]); print('  $name: %.2f ms/step  %.0f utt/s  loss %s -> %s' % (j['ms_per_step'], j['value'], j['first_loss'], j['final_loss']))
except Exception as e: print('  no JSON line', e); print(open('$D/env_$name.err').read()[-600:])
"
done
