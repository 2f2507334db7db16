#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05j
mkdir -p $O
GPU_MAX_HW_QUEUES=16 MASR_BENCH_BEAM_INPROCESS=1 timeout 900 python bench.py --no-cpu-baseline > $O/bench_q16.json 2> $O/bench_q16.err; tail -1 $O/bench_q16.err
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err
