#!/bin/bash
D=${1:-gpurun_out/r04c}; mkdir -p $D
R=$PWD
python tools/b1_ab.py 2>&1 | grep -v amdgpu > $D/b1_ab.txt; cat $D/b1_ab.txt
python -m pytest tests/test_gpu_identity.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4 > $D/tests.txt; cat $D/tests.txt
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$D/ktb -o b -- python $R/tools/b1_ab.py trace > $R/$D/ktb.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/kte -o e -- python $R/bench.py --workload efficient_b256 --steps 6 > $R/$D/kte.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find $D/ktb -name "*.db") > $D/b1_kernel_stats.txt
python profiles/summarize_rocpd.py $(find $D/kte -name "*.db") > $D/efficient_kernel_stats.txt
head -40 $D/b1_kernel_stats.txt
head -24 $D/efficient_kernel_stats.txt
find $D -name "*.db" -delete
python bench.py --workload facade > $D/facade.json 2> $D/facade.err; cat $D/facade.json
