#!/bin/bash
D=${1:-gpurun_out/r04d}; mkdir -p $D
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $D/gputests.txt; cat $D/gputests.txt
python tools/b1_ab.py 2>&1 | grep -v amdgpu > $D/b1_ab.txt; cat $D/b1_ab.txt
python bench.py --workload stream128 > $D/s128.json 2> $D/s128.err; cat $D/s128.json
MASR_BENCH_STREAMS=16 python bench.py --workload stream128 > $D/s16.json 2> $D/s16.err; cat $D/s16.json
MASR_POOL_PY=1 python bench.py --workload stream128 > $D/s128py.json 2> $D/s128py.err; cat $D/s128py.json
python bench.py --workload facade > $D/facade.json 2> $D/facade.err; cat $D/facade.json
