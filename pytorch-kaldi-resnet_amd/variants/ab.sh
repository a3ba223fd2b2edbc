export SPK_MFMA=bf16x6
for i in 1 2; do
for t in pytorch-kaldi-resnet_amd/tile_table.json pytorch-kaldi-resnet_amd/variants/tt6.json; do
SPK_TILE_TABLE=$t timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['ms_per_step'], {k:(v['ms_per_step'],v['tflops']) for k,v in d['roofline']['all_kernels'].items() if ',6>' in k and v['ms_per_step']>1})"
done; done
