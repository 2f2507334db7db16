mkdir -p gpurun_out/r06m
run() { python bench.py --workload $1 --no-cpu-baseline 2>gpurun_out/r06m/err.txt | tail -1 > gpurun_out/r06m/o.json; python -c "
import json,sys; d=json.load(open('gpurun_out/r06m/o.json')); print('$1', '$2', d['value'], d['ms_per_step'])"; }
export MASR_DEBUG_SKIP_SEARCH=1
for pp in 32 balanced 64 32,24,8 16; do
MASR_BENCH_BEAM_PASS=$pp run squeezeformer_b64_beam_sharp "no-search $pp"
done
