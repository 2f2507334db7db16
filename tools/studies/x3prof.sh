mkdir -p gpurun_out/r03c
python -m pytest tests/test_gpu_bf16x3.py -x -q 2>&1 | tail -3
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03c/kt -o x3 -- python $R/bench.py --workload bf16x3 --steps 10 > $R/gpurun_out/r03c/kt.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find gpurun_out/r03c/kt -name "*.db") > gpurun_out/r03c/x3_kernel_stats.txt 2>&1
head -30 gpurun_out/r03c/x3_kernel_stats.txt
find gpurun_out/r03c -name "*.db" -delete
tail -1 gpurun_out/r03c/kt.log
