mkdir -p gpurun_out/r06n
for w in squeezeformer_b64_beam squeezeformer_b64_beam_sharp; do python bench.py --workload $w --no-cpu-baseline 2>gpurun_out/r06n/$w.err | tail -1 > gpurun_out/r06n/$w.json; python -c "
import json,sys; d=json.load(open('gpurun_out/r06n/$w.json')); print('fresh process $w', d['value'], d['ms_per_step'])"; done
python bench.py > gpurun_out/r06n/bench.json 2> gpurun_out/r06n/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r06n/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac']); e=d['extra']; print({k:(e[k].get('ms_per_step'), e[k].get('value')) if isinstance(e[k],dict) else e[k] for k in e})"
