mkdir -p gpurun_out/r06m12; O=gpurun_out/r06m12
timeout 1200 python -m pytest tests -q -m gpu > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json | head -c 10; echo
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r06m12/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_us'])
for k, v in (j.get('extra') or {}).items():
    if isinstance(v, dict): print(k, v.get('value'), v.get('ms_per_step'))
g = j['extra']['squeezeformer_b64_greedy']['roofline']['kernels']
print([(k['avg_us'], k['frac'], k['launches']) for k in g])
PY
python tools/serve_bench.py > gpurun_out/r06m12/serving.json 2> gpurun_out/r06m12/serving.err; tail -c 900 gpurun_out/r06m12/serving.json
