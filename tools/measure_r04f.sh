#!/bin/bash
D=${1:-gpurun_out/r04f}; mkdir -p $D
python -m pytest tests/test_gpu_facade.py tests/test_gpu_regimes.py tests/test_gpu_identity.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 > $D/tests.txt; cat $D/tests.txt
MASR_AB=30:0,30:1,30:0,30:1 python tools/chunk_lat.py fused 2>&1 | tail -1 > $D/chunk_lat.txt; cat $D/chunk_lat.txt
python tools/b1_ab.py 2>&1 | grep -v amdgpu > $D/b1_ab.txt; cat $D/b1_ab.txt
MASR_BENCH_STREAMS=16 python bench.py --workload stream128 > $D/s16.json 2> $D/s16.err; python -c "import json;d=json.load(open('$D/s16.json'));print(d['value'],d['call_latency_ms'])"
python bench.py --workload stream128 > $D/s128.json 2> $D/s128.err; python -c "import json;d=json.load(open('$D/s128.json'));print(d['value'],d['call_latency_ms'])"
