#!/bin/bash
# (GPU box, after the `profile` step of run.sh in the same call) take the PMC traffic table just measured as the committed one, so that
# the bench line that follows carries `roofline.traffic` of the sources it runs (bench.py refuses a table with another csrc fingerprint)
D=${1:?run.sh passes its output directory}
cp "$D/prof/pmc_traffic.json" profiles/pmc_traffic.json && python3 -c "import json; print(json.load(open('profiles/pmc_traffic.json'))['csrc_fingerprint'])"
