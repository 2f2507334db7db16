#!/bin/bash
# One measurement pass of a build on the GPU box: GPU test suite, the default bench line, rocprofv3 kernel trace of the offline
# step and of a 16-stream StreamPool run, three PMC passes (FETCH_SIZE / WRITE_SIZE / matrix-pipe busy), summaries next to them.
# usage (from the repo root, through gpurun):  bash tools/measure_round.sh gpurun_out/r02f
D=${1:-gpurun_out/measure}; mkdir -p $D
R=$PWD
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $D/gputests.txt; cat $D/gputests.txt
python bench.py > $D/bench.json 2> $D/bench.err; tail -1 $D/bench.err
export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
B2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$D/kt -o r2 -- $B > $R/$D/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$D/pf -o f -- $B2 > $R/$D/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$D/pw -o w -- $B2 > $R/$D/pw.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d $R/$D/pm -o m -- $B2 > $R/$D/pm.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/kts -o s -- env MASR_BENCH_STREAMS=16 python $R/bench.py --workload stream128 > $R/$D/kts.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$D/ktx -o x -- python $R/bench.py --workload bf16x3 --steps 10 > $R/$D/ktx.log 2>&1
cd $R
python tools/serve_bench.py > $D/serving.json 2> $D/serving.err
for o in 0 3 5; do python tools/beam_profile.py 498 4233 300 $o 2>&1 | tail -2; done > $D/beam_profile.txt 2>&1
python tools/beam_batch_profile.py 2>&1 | grep -v amdgpu > $D/beam_batch_timeline.txt
python tools/chunk_lat.py build 2>&1 | tail -1 > $D/chunk_lat.txt
python tools/predict_long_profile.py 2>&1 | grep -v amdgpu | tail -3 > $D/predict_long.txt
python profiles/summarize_rocpd.py $(find $D/ktx -name "*.db") > $D/x3_kernel_stats.txt
python profiles/summarize_rocpd.py $(find $D/kt -name "*.db") 44 > $D/kernel_stats.txt
python profiles/summarize_rocpd.py $(find $D/kts -name "*.db") > $D/stream16_kernel_stats.txt
python profiles/summarize_pmc.py $(find $D/pf -name "*.db") $(find $D/pw -name "*.db") $D/hbm_traffic.json > $D/hbm.txt 2>&1
python profiles/summarize_mfma.py $(find $D/pm -name "*.db") $D/mfma_util.json > $D/mfma.txt 2>&1
head -14 $D/kernel_stats.txt
find $D -name "*.db" -delete
grep -c . $D/stream16_kernel_stats.txt
