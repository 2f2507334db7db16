#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05h
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/h2h_ab.py 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/h2h_ab.txt; cat $O/h2h_ab.txt
timeout 600 python -m pytest tests/test_beam_search.py tests/test_gpu_facade.py -m gpu -q 2>&1 | tail -8 > $O/pytest_beam.txt; tail -3 $O/pytest_beam.txt
for cfg in "32 0" "32 1"; do
  set -- $cfg
  echo "=== passes $1 sharp $2 (bound stage skipped on small frames)" >> $O/beam_timeline.txt
  MASR_BENCH_BEAM_PASS=$1 MASR_PROFILE_SHARP=$2 timeout 300 python tools/beam_batch_profile.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -6 >> $O/beam_timeline.txt
done
cat $O/beam_timeline.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
