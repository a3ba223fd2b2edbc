#!/bin/bash
# BASELINE configs[4] end to end at its stated size (100 k utterances x 300 frames, 9.6 GB ark): binary FV output first (with the CPU
# oracle subset), then the reference's text format on the same corpus (no oracle leg).  usage (run.sh): sh:tools/gpu/c5.sh
D=${1:-gpurun_out/c5}
mkdir -p $D
python3 tools/c5_extract.py --dir /tmp/c5 --speakers 1000 --utts-per-speaker 100 --out-format fv --oracle-subset 512 > $D/c5_fv.json 2> $D/c5_fv.err || { tail -20 $D/c5_fv.err; exit 1; }
cat $D/c5_fv.json
python3 tools/c5_extract.py --dir /tmp/c5 --speakers 1000 --utts-per-speaker 100 --out-format text --oracle-subset 0 --big-trials 0 > $D/c5_text.json 2> $D/c5_text.err || { tail -20 $D/c5_text.err; exit 1; }
cat $D/c5_text.json
rm -rf /tmp/c5
