timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_fwd" 2>&1 | tail -2
timeout -k 10 200 python tools/conv_bench.py 2>&1 | grep "3x3" | cut -c1-110
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['all_kernels'].items() if 'conv_mfma' in k and v['ms_per_step']>4})"
done
