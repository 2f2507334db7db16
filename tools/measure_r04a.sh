#!/bin/bash
# round 4, first GPU pass: facade tests + facade extras + kernel traces of the secondary workloads
D=${1:-gpurun_out/r04a}; mkdir -p $D
R=$PWD
python -m pytest tests/test_gpu_identity.py tests/test_gpu_facade.py -m gpu -q -x 2>&1 | tail -5 > $D/gputests.txt; cat $D/gputests.txt
python tools/facade_profile.py 2>&1 | grep -v amdgpu > $D/facade_profile.txt; cat $D/facade_profile.txt
python bench.py --workload facade > $D/facade.json 2> $D/facade.err; cat $D/facade.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$D/kte -o e -- python $R/bench.py --workload efficient_b256 --steps 6 > $R/$D/kte.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d $R/$D/pme -o m -- python $R/bench.py --workload efficient_b256 --steps 2 > $R/$D/pme.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find $D/kte -name "*.db") > $D/efficient_kernel_stats.txt
python profiles/summarize_mfma.py $(find $D/pme -name "*.db") $D/efficient_mfma_util.json > $D/efficient_mfma.txt 2>&1
head -40 $D/efficient_kernel_stats.txt
tail -1 $D/kte.log
find $D -name "*.db" -delete
