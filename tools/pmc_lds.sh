#!/bin/bash
# LDS / issue counters of the conv launch shapes (tools/conv_bench.py, current tiles): where do the wave cycles of the
# weight-gradient and convolution kernels go?   usage (GPU box): bash tools/pmc_lds.sh <tag>
TAG=${1:-lds}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 -L > $R/gpurun_out/counters_avail.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_$TAG --output-format csv -- python3 $R/tools/conv_bench.py --reps 2 > $R/gpurun_out/pmc_$TAG.log 2>&1
echo "rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU -d $R/gpurun_out/pmc_${TAG}b --output-format csv -- python3 $R/tools/conv_bench.py --reps 2 > $R/gpurun_out/pmc_${TAG}b.log 2>&1
echo "rc=$?"
cd $R
python3 tools/pmc_sq_summary.py gpurun_out/pmc_$TAG 12 > gpurun_out/pmc_$TAG.md 2>&1
python3 tools/pmc_sq_summary.py gpurun_out/pmc_${TAG}b 12 > gpurun_out/pmc_${TAG}b.md 2>&1
grep -c . gpurun_out/counters_avail.txt
