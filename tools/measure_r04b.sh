#!/bin/bash
D=${1:-gpurun_out/r04b}; mkdir -p $D
python tools/efficient_ab.py 2>&1 | grep -v amdgpu > $D/efficient_ab.txt; cat $D/efficient_ab.txt
python -m pytest tests/test_gpu_regimes.py -m gpu -q -x -k "efficient or batch32" 2>&1 | tail -4 > $D/regimes.txt; cat $D/regimes.txt
python bench.py --workload efficient_b256 > $D/eff.json 2> $D/eff.err; cat $D/eff.json
