#!/bin/bash
# round 5, GPU pass A: full GPU suite, the bench line with all extras, kernel trace of the Squeezeformer greedy workload
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log
tail -5 $O/bench.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_sq -o sq -- python bench.py --workload squeezeformer_b64_greedy --no-cpu-baseline > $O/sq.json 2> $O/sq.log
find $O/prof_sq -name "*kernel_stats*" | head -3
for f in $(find $O/prof_sq -name "*kernel_stats.csv" | head -1); do head -40 $f > $O/sq_kernel_stats.csv; done
rm -rf $O/prof_sq/*/*.db 2>/dev/null
du -sh $O
cat $O/pytest.txt | tail -15
